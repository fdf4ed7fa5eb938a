"""Checkpoint name maps (mmr_amd.checkpoint): OpenAI CLIP / HF CLIPModel / HF BERT -> this package's names.

CPU only.  The HF maps are checked against ``transformers`` itself (the backend the reference calls,
code/test_taiyi.py:12-24): a randomly initialised HF model's own state dict is converted, and the
oracle run on the converted weights must reproduce that model's outputs.  The OpenAI map has no
importable counterpart here (the ``clip`` package is absent): it is checked structurally, against the
tensor names and layouts of that package's ``CLIP`` module."""
import os

import pytest
import torch

import mmr_amd
from mmr_amd import checkpoint, config, synth, weights

transformers = pytest.importorskip("transformers")


def _tiny_hf_clip():
    from oracle.hf_adapter import build_hf_clip
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=3)
    return ccfg, w, build_hf_clip(ccfg, w)


def test_hf_clip_state_dict_converts_and_reproduces_hf_outputs():
    from oracle import clip_ref
    from oracle.hf_adapter import hf_image_features, hf_text_features
    ccfg, w, hf = _tiny_hf_clip()
    conv = checkpoint.from_hf_clip_state_dict(hf.state_dict())
    assert set(conv) == set(w)
    for k in w:
        assert torch.equal(conv[k], w[k]), k
    inferred = checkpoint.infer_clip_config(conv, "tiny-test")
    assert (inferred.vision, inferred.text, inferred.embed_dim) == (ccfg.vision, ccfg.text, ccfg.embed_dim)
    px = synth.synth_images(2, ccfg.vision.image_size, seed=1)
    ids = synth.synth_token_ids(3, ccfg.text.tokens, ccfg.text.vocab, seed=2)
    with torch.no_grad():
        assert torch.allclose(clip_ref.encode_image(conv, inferred.vision, px), hf_image_features(hf, px), atol=2e-5)
        assert torch.allclose(clip_ref.encode_text(conv, inferred.text, ids), hf_text_features(hf, ids), atol=2e-5)
    # and the inverse map loads back into transformers without missing/unexpected weights
    back = checkpoint.to_hf_clip_state_dict(conv)
    missing, unexpected = hf.load_state_dict(back, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing)


def test_openai_naming_round_trip_and_layout():
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=4)
    sd = checkpoint.to_openai_state_dict(w)
    d, E = ccfg.vision.width, ccfg.embed_dim
    # names and layouts of the OpenAI `clip` package's CLIP module
    assert sd["visual.proj"].shape == (d, E) and sd["text_projection"].shape == (ccfg.text.width, E)   # [d,E], x @ proj
    assert sd["visual.conv1.weight"].shape == (d, 3, ccfg.vision.patch, ccfg.vision.patch)
    assert sd["visual.transformer.resblocks.1.attn.in_proj_weight"].shape == (3 * d, d)
    assert sd["transformer.resblocks.0.mlp.c_fc.weight"].shape == (ccfg.text.mlp, ccfg.text.width)
    assert sd["positional_embedding"].shape == (ccfg.text.tokens, ccfg.text.width)
    assert len(sd) == 14 + 12 * (ccfg.vision.layers + ccfg.text.layers)
    # wrapped in DataParallel-style prefixes and fp16, as clip.load leaves them on CUDA
    wrapped = {"module." + k: v.half() for k, v in sd.items()}
    back = checkpoint.from_openai_state_dict(wrapped)
    assert set(back) == set(w)
    for k in w:
        assert back[k].dtype == torch.float32
        assert torch.equal(back[k], w[k].half().float()), k
    kind, auto = checkpoint.convert_state_dict(sd)
    assert kind == "clip" and torch.equal(auto["v.proj"], w["v.proj"])


def test_hf_bert_state_dict_converts_and_reproduces_hf_logits():
    from oracle import bert_ref
    from oracle.hf_adapter import build_hf_bert
    cfg = config.get_bert_config("tiny-bert-test")
    w = weights.make_bert_weights(cfg, seed=5)
    hf = build_hf_bert(cfg, w)
    kind, conv = checkpoint.convert_state_dict(hf.state_dict())
    assert kind == "bert" and set(conv) == set(w)
    for k in w:
        assert torch.equal(conv[k], w[k]), k
    inferred = checkpoint.infer_bert_config(conv, "tiny-bert-test")
    assert (inferred.width, inferred.layers, inferred.mlp, inferred.vocab, inferred.embed_dim) == \
           (cfg.width, cfg.layers, cfg.mlp, cfg.vocab, cfg.embed_dim)
    ids = torch.randint(1, cfg.vocab, (3, 12), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = hf(input_ids=ids).logits
        got = bert_ref.bert_logits(conv, inferred, ids)
    assert torch.allclose(got, ref, atol=2e-5)


def test_checkpoint_files_safe_loaders(tmp_path):
    ccfg, w, hf = _tiny_hf_clip()
    from safetensors.torch import save_file
    st = tmp_path / "model.safetensors"
    save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, str(st))
    cfg1, w1 = checkpoint.load_clip_checkpoint(str(st), "tiny-test")
    pt = tmp_path / "tiny.state.pt"
    torch.save({"state_dict": checkpoint.to_openai_state_dict(w)}, str(pt))
    cfg2, w2 = checkpoint.load_clip_checkpoint(str(pt))
    assert cfg1.vision == cfg2.vision == ccfg.vision
    for k in w:
        assert torch.equal(w1[k], w[k]) and torch.equal(w2[k], w[k]), k
    # a pickled module (what a TorchScript / full-model save looks like to the safe loader) is refused, not executed
    bad = tmp_path / "module.pt"
    torch.save(torch.nn.Linear(2, 2), str(bad))
    with pytest.raises(RuntimeError, match="not a plain tensor state dict"):
        checkpoint.read_state_dict(str(bad))
    with pytest.raises(KeyError, match="unrecognised checkpoint"):
        checkpoint.convert_state_dict({"foo": torch.zeros(1)})
    with pytest.raises(FileNotFoundError):
        checkpoint.read_state_dict(str(tmp_path / "absent.pt"))


@pytest.mark.gpu
def test_load_from_checkpoint_file_matches_weight_dict(tmp_path):
    """clip.load(<file>) and clip.load(name, download_root=...) run the same towers as the weight dict."""
    device = torch.device("cuda:0")
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=6)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in checkpoint.to_hf_clip_state_dict(w).items()}, str(tmp_path / "tiny-test.safetensors"))
    px = synth.synth_images(3, ccfg.vision.image_size, seed=1).to(device)
    ref_model, _ = mmr_amd.load("tiny-test", device=device, weights=w)
    ref = ref_model.encode_image(px)
    m1, _ = mmr_amd.load(str(tmp_path / "tiny-test.safetensors"), device=device)
    m2, _ = mmr_amd.load("tiny-test", device=device, download_root=str(tmp_path))
    m3, _ = mmr_amd.load("tiny-test", device=device, weights=checkpoint.to_openai_state_dict(w))
    for m in (m1, m2, m3):
        assert torch.equal(m.encode_image(px), ref)
    with pytest.raises(ValueError, match="does not match"):
        mmr_amd.load("ViT-B/32", device=device, weights=w)


# ------------------------------------------------------------------ TorchScript archives (what clip.load leaves on disk)
class _OaAttn(torch.nn.Module):
    """nn.MultiheadAttention's parameter names (what the `clip` package's resblocks hold)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = torch.nn.Parameter(torch.zeros(3 * d, d))
        self.in_proj_bias = torch.nn.Parameter(torch.zeros(3 * d))
        self.out_proj = torch.nn.Linear(d, d)

    def forward(self, x):
        return self.out_proj(x)


class _OaBlock(torch.nn.Module):
    def __init__(self, d, m):
        super().__init__()
        from collections import OrderedDict
        self.attn = _OaAttn(d)
        self.ln_1 = torch.nn.LayerNorm(d)
        self.mlp = torch.nn.Sequential(OrderedDict([("c_fc", torch.nn.Linear(d, m)), ("gelu", torch.nn.GELU()),
                                                    ("c_proj", torch.nn.Linear(m, d))]))
        self.ln_2 = torch.nn.LayerNorm(d)

    def forward(self, x):
        return x + self.mlp(self.ln_2(x + self.attn(self.ln_1(x))))


class _OaTransformer(torch.nn.Module):
    def __init__(self, d, m, layers):
        super().__init__()
        self.resblocks = torch.nn.Sequential(*[_OaBlock(d, m) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _OaVisual(torch.nn.Module):
    def __init__(self, t, E):
        super().__init__()
        d = t.width
        self.conv1 = torch.nn.Conv2d(3, d, t.patch, t.patch, bias=False)
        self.class_embedding = torch.nn.Parameter(torch.zeros(d))
        self.positional_embedding = torch.nn.Parameter(torch.zeros(t.tokens, d))
        self.ln_pre = torch.nn.LayerNorm(d)
        self.transformer = _OaTransformer(d, t.mlp, t.layers)
        self.ln_post = torch.nn.LayerNorm(d)
        self.proj = torch.nn.Parameter(torch.zeros(d, E))

    def forward(self, x):
        return self.ln_post(self.transformer(self.ln_pre(x))) @ self.proj


class _OaClip(torch.nn.Module):
    """Module tree with the tensor names of the `clip` package's CLIP class; forward is irrelevant (never run)."""

    def __init__(self, ccfg):
        super().__init__()
        t = ccfg.text
        self.visual = _OaVisual(ccfg.vision, ccfg.embed_dim)
        self.transformer = _OaTransformer(t.width, t.mlp, t.layers)
        self.token_embedding = torch.nn.Embedding(t.vocab, t.width)
        self.positional_embedding = torch.nn.Parameter(torch.zeros(t.tokens, t.width))
        self.ln_final = torch.nn.LayerNorm(t.width)
        self.text_projection = torch.nn.Parameter(torch.zeros(t.width, ccfg.embed_dim))
        self.logit_scale = torch.nn.Parameter(torch.ones([]))

    def forward(self, x):
        return self.ln_final(self.transformer(x)) @ self.text_projection * self.logit_scale


def _write_torchscript_clip(path, ccfg, w, half=True):
    m = _OaClip(ccfg)
    sd = checkpoint.to_openai_state_dict(w)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    if half:
        m = m.half()                                   # clip.load's CUDA archives hold fp16 weights
    torch.jit.save(torch.jit.script(m), str(path))


def test_torchscript_archive_is_read_without_running_it(tmp_path):
    """`clip.load("ViT-B/32")` caches OpenAI's TorchScript `ViT-B-32.pt`; that is the file a user of the
    reference has.  It must load through the allow-listed reader (no torch.jit.load, nothing executed) and convert."""
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=8)
    pt = tmp_path / "tiny-test.pt"
    _write_torchscript_clip(pt, ccfg, w)
    assert checkpoint._is_torchscript_archive(str(pt))
    with pytest.raises(Exception):
        torch.load(str(pt), weights_only=True)          # the stock safe loader refuses the archive
    sd = checkpoint.read_state_dict(str(pt))
    ref = torch.jit.load(str(pt)).state_dict()          # test-only cross-check on an archive this test wrote itself
    assert set(sd) == set(ref) and all(torch.equal(sd[k], ref[k]) for k in ref)
    assert sd["visual.conv1.weight"].dtype == torch.float16
    cfg, conv = checkpoint.load_clip_checkpoint(str(pt), "tiny-test")
    assert (cfg.vision, cfg.text) == (ccfg.vision, ccfg.text) and set(conv) == set(w)
    for k in w:
        assert torch.equal(conv[k], w[k].half().float()), k
    # a plain state-dict zip is not mistaken for an archive
    plain = tmp_path / "plain.pt"
    torch.save(checkpoint.to_openai_state_dict(w), str(plain))
    assert not checkpoint._is_torchscript_archive(str(plain))


def test_torchscript_reader_refuses_anything_but_tensors(tmp_path):
    """A data.pkl that names any global outside the allow-list (here: os.system via REDUCE) raises before anything runs;
    so do a storage view that leaves its storage and a path-escaping storage key."""
    import io
    import pickle
    import zipfile

    class Evil:
        def __reduce__(self):
            import os as _os
            return (_os.system, ("echo pwned > /tmp/mmr_pwned",))

    def archive(name, data_pkl, extra=()):
        p = tmp_path / name
        with zipfile.ZipFile(p, "w") as zf:
            zf.writestr("m/data.pkl", data_pkl)
            zf.writestr("m/constants.pkl", pickle.dumps(()))
            zf.writestr("m/code/__torch__.py", "")
            zf.writestr("m/version", "3\n")
            for k, v in extra:
                zf.writestr(k, v)
        return str(p)

    if os.path.exists("/tmp/mmr_pwned"):
        os.remove("/tmp/mmr_pwned")
    with pytest.raises(RuntimeError, match="refused"):
        checkpoint.read_state_dict(archive("evil.pt", pickle.dumps(Evil(), protocol=2)))
    assert not os.path.exists("/tmp/mmr_pwned")

    # hand-assembled pickles: module object {"w": tensor view}
    def module_with(pid_key, numel, offset, size, stride, storage_bytes):
        P = pickle
        b = io.BytesIO()
        b.write(P.PROTO + b"\x02")
        b.write(P.GLOBAL + b"__torch__\nM\n" + P.EMPTY_TUPLE + P.NEWOBJ + P.EMPTY_DICT + P.MARK)
        b.write(P.BINUNICODE + len(b"w").to_bytes(4, "little") + b"w")
        b.write(P.GLOBAL + b"torch._utils\n_rebuild_tensor_v2\n" + P.MARK)
        b.write(P.MARK + P.BINUNICODE + (7).to_bytes(4, "little") + b"storage" + P.GLOBAL + b"torch\nFloatStorage\n")
        kb = pid_key.encode()
        b.write(P.BINUNICODE + len(kb).to_bytes(4, "little") + kb + P.BINUNICODE + (3).to_bytes(4, "little") + b"cpu")
        b.write(P.BININT + numel.to_bytes(4, "little", signed=True) + P.TUPLE + P.BINPERSID)
        b.write(P.BININT + offset.to_bytes(4, "little", signed=True))
        b.write(P.MARK + b"".join(P.BININT + s.to_bytes(4, "little", signed=True) for s in size) + P.TUPLE)
        b.write(P.MARK + b"".join(P.BININT + s.to_bytes(4, "little", signed=True) for s in stride) + P.TUPLE)
        b.write(P.NEWFALSE + P.GLOBAL + b"collections\nOrderedDict\n" + P.EMPTY_TUPLE + P.REDUCE + P.TUPLE + P.REDUCE)
        b.write(P.SETITEMS + P.BUILD + P.STOP)
        return b.getvalue(), storage_bytes

    good, raw = module_with("0", 6, 0, (2, 3), (3, 1), torch.arange(6, dtype=torch.float32).numpy().tobytes())
    sd = checkpoint.read_state_dict(archive("good.pt", good, [("m/data/0", raw)]))
    assert torch.equal(sd["w"], torch.arange(6, dtype=torch.float32).view(2, 3))
    oob, raw = module_with("0", 6, 2, (2, 3), (3, 1), raw)                      # view runs past the storage
    with pytest.raises(RuntimeError, match="outside its storage"):
        checkpoint.read_state_dict(archive("oob.pt", oob, [("m/data/0", raw)]))
    esc, raw = module_with("../../etc/passwd", 6, 0, (2, 3), (3, 1), raw)       # storage key is not a plain token
    with pytest.raises(RuntimeError, match="bad storage key"):
        checkpoint.read_state_dict(archive("esc.pt", esc, [("m/data/0", raw)]))


@pytest.mark.gpu
def test_clip_load_reads_the_cached_torchscript_file(tmp_path):
    """`clip.load("ViT-B/32", download_root=d)` with d/ViT-B-32.pt in TorchScript layout (reference
    code/search_image.py:327) -- here at the tiny geometry, through the same code path."""
    device = torch.device("cuda:0")
    ccfg = mmr_amd.get_config("tiny-test")
    w = weights.make_clip_weights(ccfg, seed=8)
    _write_torchscript_clip(tmp_path / "tiny-test.pt", ccfg, w, half=False)
    px = synth.synth_images(3, ccfg.vision.image_size, seed=1).to(device)
    ids = synth.synth_token_ids(2, ccfg.text.tokens, ccfg.text.vocab, seed=2).to(device)
    ref_model, _ = mmr_amd.load("tiny-test", device=device, weights=w)
    m, _ = mmr_amd.load("tiny-test", device=device, download_root=str(tmp_path))
    assert not m.synthetic_weights
    assert torch.equal(m.encode_image(px), ref_model.encode_image(px))
    assert torch.equal(m.encode_text(ids), ref_model.encode_text(ids))
