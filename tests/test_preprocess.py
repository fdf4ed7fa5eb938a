"""Preprocess: host coefficient tables vs Pillow on the CPU; HIP kernel vs Pillow on the GPU."""
import numpy as np
import pytest
import torch

SIZES = [(37, 53), (224, 224), (300, 200), (200, 300), (1000, 751), (97, 640), (224, 225), (64, 64), (231, 500)]


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if seed % 2:                      # smooth-ish content as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        base = ((np.sin(yy / 7.0)[..., None] * 90 + np.cos(xx / 11.0)[..., None] * 90 + 128
                 + rng.integers(-20, 20, (h, w, 3)))).clip(0, 255).astype(np.uint8)
    return base


def _apply_tables(img, n_px):
    """Pure-numpy application of the product's tables (int32 math), to check them without a GPU."""
    from mmr_amd import preprocess as P
    h, w = img.shape[:2]
    nh, nw, top, left = P.resize_geometry(h, w, n_px)
    hb, hc, _ = P.resample_tables(w, nw, left, n_px)
    vb, vc, _ = P.resample_tables(h, nh, top, n_px)
    half = 1 << (P.PRECISION_BITS - 1)
    src = img.astype(np.int64)
    tmp = np.zeros((h, n_px, 3), np.int64)
    for x in range(n_px):
        x0, n = hb[x]
        acc = half + (src[:, x0:x0 + n, :] * hc[x, :n, None].astype(np.int64)).sum(axis=1)
        tmp[:, x, :] = np.clip(acc >> P.PRECISION_BITS, 0, 255)
    out = np.zeros((n_px, n_px, 3), np.uint8)
    for y in range(n_px):
        y0, n = vb[y]
        acc = half + (tmp[y0:y0 + n] * vc[y, :n, None, None].astype(np.int64)).sum(axis=0)
        out[y] = np.clip(acc >> P.PRECISION_BITS, 0, 255)
    return out


@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("n_px", [224, 64])
def test_tables_reproduce_pillow_bit_exact(hw, n_px):
    from oracle import preprocess_ref
    img = _img(hw[0], hw[1], seed=hw[0] + hw[1])
    assert np.array_equal(_apply_tables(img, n_px), preprocess_ref.preprocess_u8(img, n_px))


def test_geometry_matches_oracle():
    from mmr_amd import preprocess as P
    from oracle import preprocess_ref
    for h, w in SIZES + [(225, 224), (449, 224), (224, 1000)]:
        for n in (224, 336):
            assert P.resize_geometry(h, w, n) == preprocess_ref.geometry(h, w, n)


@pytest.mark.gpu
@pytest.mark.parametrize("hw", SIZES)
def test_hip_preprocess_bit_exact_vs_pillow(device, hw):
    from mmr_amd import preprocess as P
    from oracle import preprocess_ref
    img = _img(hw[0], hw[1], seed=hw[0] * 3 + hw[1])
    for n_px in (224, 64):
        out, u8 = P.preprocess_image(torch.from_numpy(img).to(device), n_px, return_u8=True)
        assert np.array_equal(u8.cpu().numpy(), preprocess_ref.preprocess_u8(img, n_px))     # byte-exact resize+crop
        ref = preprocess_ref.preprocess(img, n_px)
        assert torch.equal(out.cpu(), ref)                                                   # and fp32-exact normalise
    ob = P.preprocess_image(torch.from_numpy(img).to(device), 224, out_dtype=torch.bfloat16)
    assert torch.equal(ob.cpu(), preprocess_ref.preprocess(img, 224).bfloat16())


@pytest.mark.gpu
def test_preprocess_feeds_encoder(device):
    import mmr_amd
    from mmr_amd import preprocess as P
    from oracle import preprocess_ref
    model, _ = mmr_amd.load("tiny-test", device=device)
    imgs = [_img(80 + 13 * i, 120 - 7 * i, seed=i) for i in range(5)]
    batch = P.preprocess_batch([torch.from_numpy(a).to(device) for a in imgs], model.input_resolution)
    ref = torch.stack([preprocess_ref.preprocess(a, model.input_resolution) for a in imgs])
    assert torch.equal(batch.cpu(), ref)                      # one launch pair for 5 differently sized images
    big = [_img(h, w, seed=h) for h, w in SIZES]
    b224 = P.preprocess_batch([torch.from_numpy(a).to(device) for a in big], 224, out_dtype=torch.bfloat16)
    assert torch.equal(b224.cpu(), torch.stack([preprocess_ref.preprocess(a, 224) for a in big]).bfloat16())
    f = model.encode_image(batch)
    assert f.shape == (5, model.cfg.embed_dim) and torch.isfinite(f).all()
    with pytest.raises(ValueError):
        P.preprocess_image(torch.zeros(4, 4, 3, device=device))
    with pytest.raises(RuntimeError):
        P.preprocess_image(torch.zeros(4, 4, 3, dtype=torch.uint8))


@pytest.mark.gpu
def test_overlapped_gallery_build_equals_serial_and_pillow(device):
    """the reference's gallery loop preprocess -> encode_image -> normalise -> keep the row
    (code/search_image.py:153-158) with the preprocess of batch i+1 on a side stream under the encode of batch i:
    bit-identical to the serial schedule, and to Pillow-exact per-image preprocessing + encode_image; ragged last batch."""
    import mmr_amd
    from mmr_amd import gallery, preprocess as P
    from oracle import preprocess_ref
    model, _ = mmr_amd.load("tiny-test", device=device)
    S, E = model.input_resolution, model.cfg.embed_dim
    H, W = 90, 130
    g = torch.Generator().manual_seed(5)
    raws = [torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, generator=g) for n in (6, 6, 6, 6, 3)]   # ragged tail
    dev_raws = [r.to(device) for r in raws]
    total = sum(r.shape[0] for r in raws)
    ovl = gallery.build_gallery_overlapped(model, iter(dev_raws), total=total, overlap=True).clone()
    ser = gallery.build_gallery_overlapped(model, iter(dev_raws), total=total, overlap=False, lanes=1).clone()
    assert ovl.shape == (total, E) and ovl.dtype == torch.bfloat16
    assert torch.equal(ovl, ser)
    # the default schedule: two (three) batches in flight on their own streams, pixel buffers and model workspaces; the
    # batches come from a generator that makes each one on the caller's stream just before it is consumed
    def fresh():
        for r in raws:
            yield r.to(device)
    for lanes in (2, 3):
        two = gallery.build_gallery_overlapped(model, fresh(), total=total, lanes=lanes).clone()
        assert torch.equal(two, ser), lanes
    # reference composition: Pillow-exact pixels (oracle) rounded to bf16 -> encode_image(normalize=True) in bf16
    model.bfloat16()
    px = torch.stack([preprocess_ref.preprocess(im.numpy(), S) for r in raws for im in r]).bfloat16().to(device)
    ref = torch.cat([model.encode_image(px[i:i + 6], normalize=True) for i in range(0, total, 6)])
    model.float()
    assert torch.equal(ovl, ref)
    assert model.dtype == torch.float32                       # the caller's dtype setting is restored
    # the uniform-batch preprocessor alone == the per-image path
    pre = P.UniformBatchPreprocessor(6, H, W, S, out_dtype=torch.float32, device=device, slots=1)
    assert torch.equal(pre(dev_raws[0]).cpu(), torch.stack([preprocess_ref.preprocess(im.numpy(), S) for im in raws[0]]))
    assert gallery.build_gallery_overlapped(model, iter([]), total=0).shape == (0, E)
    with pytest.raises(ValueError):
        gallery.build_gallery_overlapped(model, iter(dev_raws))                  # no total / gallery to write into
    with pytest.raises(ValueError):
        gallery.build_gallery_overlapped(model, iter(dev_raws), total=7)         # too small
    with pytest.raises(ValueError):
        model.encode_image(px[:2].float(), out=torch.empty(3, E, device=device))


@pytest.mark.gpu
def test_fast_preprocess_paths_edge_cases(device):
    """Round 3's kernels read 12 / 16 bytes at a time where round 2 read single bytes: the cases where that could go wrong.
      * images whose bytes start at an ODD address (a slice of a flat buffer) and END at the last byte of their allocation
        (the window of the last row's rightmost column must not read past it: the bounded byte-wise path);
      * very wide images (fewer rows fit the LDS stage; > 21 845 pixels wide: the non-staged 12-byte path);
      * an output size that is not a multiple of 4 (the generic vertical pass), tiny sources (up-scaling: 1-2 taps);
      * mixed batches (different widths, one launch pair) and the UniformBatchPreprocessor on the same images."""
    from mmr_amd import preprocess as P
    from oracle import preprocess_ref
    cases = [(40, 33), (50, 7000), (8, 22000), (5, 5), (231, 501), (33, 40)]
    for h, w in cases:
        img = _img(h, w, seed=h * 7 + w)
        n = h * w * 3
        flat = torch.zeros(n + 1, dtype=torch.uint8, device=device)          # image = the LAST n bytes, starting at offset 1
        flat[1:] = torch.from_numpy(img).reshape(-1).to(device)
        view = flat[1:].view(h, w, 3)
        assert view.data_ptr() % 2 == 1 or view.data_ptr() % 16 != 0
        for n_px in (224, 64, 30):                                           # 30: not a multiple of 4
            out, u8 = P.preprocess_image(view, n_px, return_u8=True)
            assert np.array_equal(u8.cpu().numpy(), preprocess_ref.preprocess_u8(img, n_px)), (h, w, n_px)
            assert torch.equal(out.cpu(), preprocess_ref.preprocess(img, n_px)), (h, w, n_px)
    imgs = [_img(h, w, seed=3 * h + w) for h, w in [(60, 90), (90, 60), (50, 7000), (224, 224)]]
    dev_imgs = [torch.from_numpy(a).to(device) for a in imgs]
    for n_px in (224, 30):
        got = P.preprocess_batch(dev_imgs, n_px, out_dtype=torch.float32)
        ref = torch.stack([preprocess_ref.preprocess(a, n_px) for a in imgs])
        assert torch.equal(got.cpu(), ref), n_px
    same = np.stack([_img(77, 131, seed=s) for s in range(5)])
    pre = P.UniformBatchPreprocessor(8, 77, 131, 64, out_dtype=torch.float32, device=device, slots=2)
    for slot in (0, 1):
        got = pre(torch.from_numpy(same).to(device), slot)
        assert torch.equal(got.cpu(), torch.stack([preprocess_ref.preprocess(a, 64) for a in same]))
    with pytest.raises(ValueError):
        pre(torch.zeros(9, 77, 131, 3, dtype=torch.uint8, device=device))
    with pytest.raises(ValueError):
        pre(torch.zeros(2, 77, 130, 3, dtype=torch.uint8, device=device))
