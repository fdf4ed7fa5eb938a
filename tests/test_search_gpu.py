"""GPU parity: HIP similarity / top-k (through the C ABI) vs the CPU oracle and the golden fixtures."""
import os

import numpy as np
import pytest
import torch

from mmr_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(device):
    from mmr_amd import search
    return search


@pytest.fixture(scope="module")
def oracle():
    from oracle import search_ref
    return search_ref


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, "search.npz"))


@pytest.mark.parametrize("N", [1000, 10000])
@pytest.mark.parametrize("Q", [1, 7, 128])
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_topk_matches_golden(S, device, golden_dir, N, Q, tag):
    g = _gold(golden_dir)
    key = f"N{N}_Q{Q}_{tag}"
    gal = synth.synth_unit_rows(N, 512, seed=int(g["meta"][0]))
    q = synth.synth_unit_rows(Q, 512, seed=int(g[key + "_qseed"]))
    if tag == "bf16":
        gal, q = gal.bfloat16(), q.bfloat16()
    vals, idx, d64, status = S.cosine_topk(q.to(device), gal.to(device), 10, scale=100.0,
                                           return_dot64=True, return_status=True)
    assert np.array_equal(idx.cpu().numpy().astype(np.int32), g[key + "_idx"])       # bit-exact ranks
    assert np.array_equal(d64.cpu().numpy(), g[key + "_dot64"])                      # same fp64 order
    assert np.array_equal(vals.cpu().numpy(), g[key + "_score"])
    assert int(status.sum()) == 0, "tie-free fixture should certify on the MFMA fast path (bf16 and fp32 scans)"


def test_tie_rule_lowest_index_first(S, device, golden_dir):
    g = _gold(golden_dir)
    gal = synth.synth_unit_rows(2048, 512, seed=int(g["meta"][1])).bfloat16()
    q = synth.synth_unit_rows(5, 512, seed=int(g["meta"][2])).bfloat16()
    gal[1500:1520] = gal[7]
    gal[40] = gal[900]
    q[0] = gal[7]
    q[1] = gal[900]
    vals, idx, d64 = S.cosine_topk(q.to(device), gal.to(device), 10, scale=100.0, return_dot64=True)
    assert np.array_equal(idx.cpu().numpy().astype(np.int32), g["ties_idx"])
    assert np.array_equal(d64.cpu().numpy(), g["ties_dot64"])
    assert idx[0].tolist() == [7] + list(range(1500, 1509))


@pytest.mark.parametrize("E", [128, 256, 512, 768, 1024])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_topk_vs_oracle_shapes(S, oracle, device, E, dtype):
    # ragged sizes: N not a multiple of the 32-row tile, Q not a multiple of 32, several k
    for N, Q, k in [(1, 1, 1), (31, 3, 5), (33, 2, 10), (4097, 37, 10), (20011, 70, 20), (3000, 5, 30)]:
        gal = synth.synth_unit_rows(N, E, seed=100 + N).to(dtype)
        q = synth.synth_unit_rows(Q, E, seed=200 + Q).to(dtype)
        vals, idx, d64 = S.cosine_topk(q.to(device), gal.to(device), k, scale=100.0, return_dot64=True)
        oi, os_, od = oracle.cosine_topk(q, gal, k, scale=100.0)
        assert np.array_equal(idx.cpu().numpy(), oi), (E, dtype, N, Q, k)
        assert np.array_equal(d64.cpu().numpy(), od)
        assert np.array_equal(vals.cpu().numpy(), os_)


def test_topk_more_than_256_queries_and_k_gt_n(S, oracle, device):
    gal = synth.synth_unit_rows(5000, 512, seed=7).bfloat16()
    q = synth.synth_unit_rows(300, 512, seed=8).bfloat16()
    vals, idx = S.cosine_topk(q.to(device), gal.to(device), 10)
    oi, os_, _ = oracle.cosine_topk(q, gal, 10)
    assert np.array_equal(idx.cpu().numpy(), oi)
    small = gal[:4]
    vals, idx = S.cosine_topk(q[:3].to(device), small.to(device), 8)
    oi, os_, _ = oracle.cosine_topk(q[:3], small, 8)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert (idx[:, 4:] == -1).all() and torch.isinf(vals[:, 4:]).all()
    # empty gallery
    vals, idx = S.cosine_topk(q[:2].to(device), gal[:0].to(device), 3)
    assert (idx == -1).all() and torch.isinf(vals).all()


def test_uncertified_queries_fall_back_to_exact(S, oracle, device):
    # 40 near-identical rows within bf16 rounding of each other straddle the top-k boundary:
    # the certificate must reject the fast path and the exhaustive path must still be exact.
    torch.manual_seed(0)
    E = 512
    gal = synth.synth_unit_rows(8192, E, seed=21).bfloat16()
    base = gal[5].clone()
    rows = torch.randperm(8192)[:60]
    for j, r in enumerate(rows.tolist()):
        v = base.clone()
        bits = v.view(torch.int16)
        bits[j] += (j % 5) - 2          # a few bf16 ulps on one component: dot moves by ~1e-5
        gal[r] = v
    q = base.unsqueeze(0).repeat(2, 1)
    q[1] = synth.synth_unit_rows(1, E, seed=22).bfloat16()[0]
    vals, idx, d64, status = S.cosine_topk(q.to(device), gal.to(device), 10, return_dot64=True, return_status=True)
    oi, os_, od = oracle.cosine_topk(q, gal, 10)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(d64.cpu().numpy(), od)
    assert int(status[0]) == 1          # crowded boundary -> exact path
    assert int(status[1]) == 0          # ordinary query -> fast path


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_unnormalised_gallery_certificate_is_sized_from_the_data(S, oracle, device, dtype):
    """Raw encode_image rows are NOT unit-norm (|f| ~ 25 for the seeded ViT-B/32).  The certificate's margin must come
    from the gallery's real row norms, not from a caller's promise: with rows scaled 5-30x and a crowded top-k
    boundary (the near-tie construction of test_uncertified_queries_fall_back_to_exact) the result stays bit-exact
    and the crowded query is sent down the exhaustive path -- through cosine_topk without a bound (measured per call),
    through GalleryIndex (measured once) and even when the caller UNDERSTATES the bound to the index."""
    E, N = 512, 8192
    g = torch.Generator().manual_seed(5)
    scales = 5.0 + 25.0 * torch.rand(N, 1, generator=g)
    scales[5] = 20.0                                    # the duplicated row scores ~20: above any unrelated row (<= ~6)
    gal = (synth.synth_unit_rows(N, E, seed=21) * scales).to(dtype)
    base = gal[5].clone()
    rows = torch.randperm(N, generator=g)[:60]
    for j, r in enumerate(rows.tolist()):
        v = base.clone()
        if dtype == torch.bfloat16:
            v.view(torch.int16)[j] += (j % 5) - 2          # a few ulps on one component
        else:
            v.view(torch.int32)[j] += ((j % 5) - 2) * 3
        gal[r] = v
    q = torch.stack([base.float() / base.float().norm(), synth.synth_unit_rows(1, E, seed=22)[0]]).to(dtype)
    oi, os_, od = oracle.cosine_topk(q, gal, 10)
    gd, qd = gal.to(device), q.to(device)
    nb = S.gallery_norm_bound(gd)
    true_max = gal.double().norm(dim=-1).max().item()
    assert true_max <= float(nb) <= true_max * 1.001
    for how in ("per-call", "index", "index-understated"):
        if how == "per-call":
            vals, idx, d64, status = S.cosine_topk(qd, gd, 10, return_dot64=True, return_status=True)
        else:
            index = S.GalleryIndex(gd, norm_bound=1.0 if how == "index-understated" else None)
            vals, idx, d64, status = index.search(qd, 10, return_dot64=True, return_status=True)
        assert np.array_equal(idx.cpu().numpy(), oi), how
        assert np.array_equal(d64.cpu().numpy(), od), how
        assert int(status[0]) == 1 and int(status[1]) == 0, (how, status.tolist())
    # an honest explicit bound is used as given (no measuring pass) and gives the same answer
    vals, idx, status = S.cosine_topk(qd, gd, 10, gallery_norm_bound=31.0, return_status=True)
    assert np.array_equal(idx.cpu().numpy(), oi) and int(status[0]) == 1
    with pytest.raises(ValueError):
        S.cosine_topk(qd, gd, 10, gallery_norm_bound=float("inf"))


def test_index_one_dimensional_query_shapes(S, oracle, device):
    """A single 1-D query (the reference's ref_feature form): every output is squeezed the same way."""
    gal = synth.synth_unit_rows(3000, 512, seed=61).bfloat16()
    q = synth.synth_unit_rows(1, 512, seed=62).bfloat16()[0]
    oi, os_, od = oracle.cosine_topk(q.unsqueeze(0), gal, 10)
    index = S.GalleryIndex(gal.to(device))
    score, idx, d64 = index.search(q.to(device), 10, return_dot64=True)
    assert score.shape == idx.shape == d64.shape == (10,)
    assert np.array_equal(idx.cpu().numpy(), oi[0]) and np.array_equal(d64.cpu().numpy(), od[0])
    sh = S.ShardedGalleryIndex(gal.to(device))            # no process group: one local shard, same code path
    s2, i2 = sh.search(q.to(device), 10)
    assert i2.shape == (10,) and np.array_equal(i2.cpu().numpy(), oi[0])
    outs = sh.search_pipelined([q.to(device).unsqueeze(0)] * 3, 10)
    assert all(np.array_equal(o[1].cpu().numpy(), oi) for o in outs)


def test_fp32_gallery_fast_path_and_fallback(S, oracle, device):
    """fp32 galleries (the dtype of the reference's feature caches) take the fp32-MFMA scan; crowded boundaries
    still fall back to the exhaustive path and stay exact."""
    gal = synth.synth_unit_rows(70001, 512, seed=51)
    q = synth.synth_unit_rows(130, 512, seed=52)              # > 128 queries: two scan passes
    dup = list(range(1000, 41000, 1000))                      # 40 exact duplicates spread over 40 different tiles
    gal[dup] = gal[123].clone()
    q[0] = gal[123].clone()
    vals, idx, d64, status = S.cosine_topk(q.to(device), gal.to(device), 10, scale=100.0, return_dot64=True,
                                           return_status=True)
    oi, os_, od = oracle.cosine_topk(q, gal, 10, scale=100.0)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d64.cpu().numpy(), od)
    assert np.array_equal(vals.cpu().numpy(), os_)
    assert idx[0].tolist() == [123] + dup[:9]                  # ties -> lowest row ids
    assert int(status[0]) == 1 and int(status[1:].sum()) == 0   # more tied tiles than KS: the only fallback


def test_similarity_and_l2norm(S, oracle, device):
    gal = synth.synth_unit_rows(777, 512, seed=3) * 3.0
    ref = synth.synth_unit_rows(4, 512, seed=4) * 0.5       # un-normalised query, like outlier_filter's mean
    sim = S.similarity(gal.to(device), ref.to(device), 100.0)      # [N,Q]
    osim = oracle.similarity(ref, gal, 100.0)                      # [Q,N]
    assert sim.shape == (777, 4)
    assert np.array_equal(sim.t().cpu().numpy(), osim)
    one = S.similarity(gal.to(device), ref[0].to(device), 100.0)
    assert one.shape == (777,)
    assert np.array_equal(one.cpu().numpy(), osim[0])
    # matches the reference's torch expression to fp32 rounding
    assert torch.allclose(sim.cpu(), 100.0 * gal @ ref.t(), atol=2e-3)
    for dt, tol in ((torch.float32, 2e-6), (torch.bfloat16, 1e-2)):
        x = (torch.randn(33, 512) * 5).to(dt)
        y = S.l2_normalize(x.to(device))
        ref_n = oracle.l2norm_rows(x)
        assert np.abs(y.float().cpu().numpy() - ref_n).max() <= tol
    # in-place form used by callers: feats /= feats.norm(...)
    x = torch.randn(5, 512, device=device)
    y = S.l2_normalize(x, inplace=True)
    assert y.data_ptr() == x.data_ptr() and torch.allclose(x.norm(dim=-1), torch.ones(5, device=device), atol=1e-5)


def test_merge_equals_unsharded(S, oracle, device):
    gal = synth.synth_unit_rows(6000, 512, seed=31).bfloat16()
    q = synth.synth_unit_rows(9, 512, seed=32).bfloat16()
    bounds = [0, 1000, 1001, 4000, 6000]
    parts_i, parts_d = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        v, i, d = S.cosine_topk(q.to(device), gal[a:b].to(device), 10, return_dot64=True)
        parts_i.append(torch.where(i >= 0, i + a, i))
        parts_d.append(d)
    v, i, d = S.merge_topk(torch.stack(parts_i), torch.stack(parts_d), 1.0)
    oi, os_, od = oracle.cosine_topk(q, gal, 10)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od)


def test_error_paths(S, device):
    from mmr_amd._lib import MMRError
    gal = torch.randn(10, 512, device=device)
    with pytest.raises(MMRError):
        S.cosine_topk(gal[:1], gal, 0)                  # k < 1
    with pytest.raises(MMRError):
        S.cosine_topk(gal[:1], gal, 10, scale=-1.0)     # scale must be positive
    with pytest.raises(MMRError):
        S.cosine_topk(torch.randn(1, 96, device=device), torch.randn(4, 96, device=device), 1)  # E unsupported
    with pytest.raises(ValueError):
        S.cosine_topk(torch.randn(1, 256, device=device), gal, 1)
    with pytest.raises(RuntimeError):
        S.cosine_topk(torch.randn(1, 512), torch.randn(4, 512), 1)  # CPU tensors: no CPU path


@pytest.mark.slow
def test_selection_when_the_best_tasks_crowd_into_few_lanes(S, oracle, device):
    """select_kernel ranks its candidates (505 scan tasks at 1M rows, candidate i in lane i % 64) from a pivot = the
    (KS+1)-th largest per-lane maximum; when the strong candidates sit in fewer lanes than that, more than 64 candidates
    pass the pivot and the bitwise search for the exact cut takes over.  64 strong rows, one in every task t with
    t % 64 < 8, graded so that the top-10 is unambiguous; a second query is plain random."""
    N, E, k = 1_000_000, 512, 10
    gal = synth.synth_unit_rows(N, E, seed=13, dtype=torch.bfloat16)
    q = synth.synth_unit_rows(2, E, seed=14, dtype=torch.bfloat16)
    rows_per_task = 62 * 32                                  # make_plan(): 31 250 tiles -> 62 tiles per task, 505 tasks
    planted = []
    g = torch.Generator().manual_seed(15)
    for j, t in enumerate(t for t in range(505) if t % 64 < 8):
        r = t * rows_per_task + 5 + (j % 7) * 32
        noise = torch.randn(E, generator=g)
        q0 = q[0].float() / q[0].float().norm()
        noise = noise - (noise @ q0) * q0                    # orthogonal to the query: the cosine is set by w alone
        noise = noise / noise.norm()
        w = 0.45 + 0.006 * j                                 # cosines 0.91 .. 0.74, about 0.003 apart
        v = q[0].float() * (1 - w * w) ** 0.5 + noise * w
        gal[r] = (v / v.norm()).bfloat16()
        planted.append(r)
    index = S.GalleryIndex(gal.to(device))
    vals, idx, d64, status = index.search(q.to(device), k, return_dot64=True, return_status=True)
    oi, _, od = oracle.cosine_topk(q, gal, k)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d64.cpu().numpy(), od)
    assert idx[0].tolist() == planted[:k]
    assert int(status.sum()) == 0                            # nothing near the cut: both queries certify on the fast path


@pytest.mark.slow
@pytest.mark.parametrize("N,E,Q", [(1_000_000, 512, 256),      # BASELINE configs[1]/[2]: the 1M x 512 gallery
                                   (1_250_000, 512, 256),      # configs[3]: one GPU's shard of the 10M gallery
                                   (1_000_000, 768, 128)])     # configs[4]: the E = 768 (ViT-L/14) shard
def test_full_size_gallery_exact(S, oracle, device, N, E, Q):
    """BASELINE-size shards, k=10 -- indices bit-exact vs the oracle on a query sample, plus size-independent
    properties on all queries."""
    k = 10
    gal = synth.synth_unit_rows(N, E, seed=3, dtype=torch.bfloat16)
    q = synth.synth_unit_rows(Q, E, seed=4, dtype=torch.bfloat16)
    gd = gal.to(device)
    index = S.GalleryIndex(gd)
    vals, idx, d64, status = index.search(q.to(device), k, return_dot64=True, return_status=True)
    idx_c, d_c = idx.cpu(), d64.cpu()
    # properties: sorted by (-dot,+idx), ids unique and in range, dots re-computable from the rows
    assert (d_c[:, :-1] >= d_c[:, 1:]).all()
    assert ((idx_c >= 0) & (idx_c < N)).all()
    assert all(len(set(r.tolist())) == k for r in idx_c)
    rows = gal[idx_c.reshape(-1)].double().reshape(Q, k, E)
    re = (rows * q.double().unsqueeze(1)).sum(-1)
    assert (re - d_c).abs().max() < 1e-12
    # idempotence: searching the gathered top-k rows alone returns the same order
    sub = gd[idx[0]]
    _, i2 = S.cosine_topk(q[:1].to(device), sub, k)
    assert i2[0].tolist() == list(range(k))
    # the (k+1)-th best bounds everything that was left out: top-(k+1) extends top-k
    _, i11 = index.search(q[:8].to(device), k + 1)
    assert torch.equal(i11[:, :k].cpu(), idx_c[:8])
    # oracle on a sample of queries (exhaustive fp64 over all rows on the CPU)
    sample = [0, 17, Q - 1]
    oi, _, od = oracle.cosine_topk(q[sample], gal, k)
    assert np.array_equal(idx_c[sample].numpy(), oi)
    assert np.array_equal(d_c[sample].numpy(), od)
    print("status (queries on exhaustive path):", int(status.sum()))
    assert int(status.sum()) <= 2, "the certificate should pass (almost) every query of a random unit gallery"
    # a 4-way split of the same shard + merge == the unsplit result (the sharded path's arithmetic at this size)
    cuts = [0, N // 4, N // 2, 3 * N // 4, N]
    pi, pd = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        _, i, d = S.cosine_topk(q.to(device), gd[a:b], k, return_dot64=True)
        pi.append(torch.where(i >= 0, i + a, i)); pd.append(d)
    _, mi, md = S.merge_topk(torch.stack(pi), torch.stack(pd), 1.0)
    assert torch.equal(mi.cpu(), idx_c) and torch.equal(md.cpu(), d_c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tip_adapter_logits_matches_reference_expression(S, device, dtype):
    """Row 8f-3: the fused op vs the reference's own expression (code/main_custom.py:111,124-127) in fp32."""
    N, E, C, shots = 333, 512, 6, 16
    f = synth.synth_unit_rows(N, E, seed=41).to(dtype)
    W = synth.synth_unit_rows(C, E, seed=42).t().contiguous().to(dtype)               # clip_weights [E,C]
    Kc = synth.synth_unit_rows(C * shots, E, seed=43).t().contiguous().to(dtype)      # cache_keys  [E,S]
    V = torch.nn.functional.one_hot(torch.arange(C * shots) % C, C).float()           # cache_values [S,C]
    alpha, beta = 1.17, 5.5
    ff, Wf, Kf = f.float(), W.float(), Kc.float()
    clip_logits = 100.0 * ff @ Wf
    affinity = ff @ Kf
    cache_logits = ((-1) * (beta - beta * affinity)).exp() @ V * 10
    ref = clip_logits + cache_logits * alpha
    tip, clip = S.tip_adapter_logits(f.to(device), W.to(device), Kc.to(device), V.to(device), alpha, beta,
                                     return_clip_logits=True)
    assert tip.shape == (N, C) and tip.dtype == torch.float32
    assert (clip.cpu() - clip_logits).abs().max().item() <= 2e-3          # |logit| <= 100: 2e-5 relative
    assert (tip.cpu() - ref).abs().max().item() <= 2e-3
    # the ranking the reference derives from it (cls_acc, code/utils.py:17) is identical
    assert torch.equal(tip.cpu().topk(1, 1, True, True)[1], ref.topk(1, 1, True, True)[1])
    with pytest.raises(ValueError):
        S.tip_adapter_logits(f.to(device), W[:100].to(device), Kc.to(device), V.to(device), alpha, beta)


def test_selection_paths_for_many_tasks(device):
    """Task counts above 1024 (register selection over 4 waves) and above 4096 (global sweep per round) are
    reached by very large shards; MMR_SEARCH_TPT=1 reaches them with a small gallery (separate process: the
    hook is read once per process)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        from mmr_amd import search, synth
        from oracle import search_ref
        dev = torch.device("cuda:0")
        for N in (50_000, 150_000):          # 1563 and 4688 one-tile tasks
            gal = synth.synth_unit_rows(N, 256, seed=5).bfloat16()
            q = synth.synth_unit_rows(5, 256, seed=6).bfloat16()
            v, i, d, st = search.cosine_topk(q.to(dev), gal.to(dev), 10, return_dot64=True, return_status=True)
            oi, _, od = search_ref.cosine_topk(q, gal, 10)
            assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od), N
            assert int(st.sum()) == 0
        print("ok")
    """ % (str(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))),))
    env = dict(__import__("os").environ, MMR_SEARCH_TPT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_sharded_index_over_rccl(S, oracle, device):
    """ShardedGalleryIndex through the real collective backend ("nccl" = RCCL) on this GPU: a one-rank group exercises
    the all-gather + merge code path the multi-GPU bench takes (more ranks on one card are refused by RCCL; the
    world_size-2 case runs over gloo in tests/test_host_logic.py)."""
    import socket

    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        gal = synth.synth_unit_rows(20000, 512, seed=31).bfloat16()
        q = synth.synth_unit_rows(16, 512, seed=32).bfloat16()
        index = S.ShardedGalleryIndex(gal.to(device))
        score, idx = index.search(q.to(device), 10, 100.0)
        oi, os_, _ = oracle.cosine_topk(q, gal, 10, 100.0)
        assert idx.dtype == torch.int64
        assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(score.cpu().numpy(), os_)
        s1, i1 = index.search(q[3].to(device), 10, 100.0)          # 1-D query through the collective
        assert i1.shape == (10,) and np.array_equal(i1.cpu().numpy(), oi[3])
        outs = index.search_pipelined([q[:5].to(device), q[5:].to(device)], 10, 100.0)   # async all-gathers in flight
        assert np.array_equal(torch.cat([o[1] for o in outs]).cpu().numpy(), oi)
    finally:
        dist.destroy_process_group()


def test_c_abi_rccl_allgather_of_topk(S, oracle, device):
    """The non-Python host's multi-GPU leg (include/mmr.h: mmr_comm_* + mmr_allgather_topk[_packed]): a one-rank RCCL
    communicator built through the C ABI gathers this rank's (global id, fp64 dot) lists and mmr_topk_merge ranks them --
    the same result as the local search.  (More ranks need more GPUs; the exchange itself is rank-count agnostic.)"""
    import ctypes
    from mmr_amd import _lib
    L = _lib.lib()
    uid = ctypes.create_string_buffer(128)
    _lib.check(L.mmr_comm_unique_id(uid))
    comm = ctypes.c_void_p()
    _lib.check(L.mmr_comm_init(0, 1, uid, ctypes.byref(comm)))
    try:
        gal = synth.synth_unit_rows(9000, 512, seed=71).bfloat16()
        q = synth.synth_unit_rows(12, 512, seed=72).bfloat16()
        _, lidx, ldot = S.cosine_topk(q.to(device), gal.to(device), 10, return_dot64=True)
        offset = 1_000_000
        gidx = (lidx + offset).contiguous()
        ip = torch.empty(1, 12, 10, dtype=torch.int64, device=device)
        dp = torch.empty(1, 12, 10, dtype=torch.float64, device=device)
        _lib.check(L.mmr_allgather_topk(comm, gidx.data_ptr(), ldot.data_ptr(), 12, 10, ip.data_ptr(), dp.data_ptr(),
                                        _lib.stream_ptr(device)))
        torch.cuda.synchronize(device)
        assert torch.equal(ip[0], gidx) and torch.equal(dp[0], ldot)
        score, idx, d64 = S.merge_topk(ip, dp, 100.0)
        oi, os_, od = oracle.cosine_topk(q, gal, 10, 100.0)
        assert np.array_equal((idx - offset).cpu().numpy(), oi) and np.array_equal(d64.cpu().numpy(), od)
        assert np.array_equal(score.cpu().numpy(), os_)
        assert L.mmr_allgather_topk(None, gidx.data_ptr(), ldot.data_ptr(), 12, 10, ip.data_ptr(), dp.data_ptr(), 0) == -22
        # the packed form -- the sequence include/mmr.h and INTEGRATION.md give, and the exchange the Python index issues:
        # mmr_cosine_topk_ex (int32 local ids) -> mmr_topk_pack -> ONE all-gather -> mmr_topk_merge_packed
        gd, qd = gal.to(device), q.to(device)
        ws = torch.empty(max(256, L.mmr_search_workspace_bytes(9000, 512, 12, 10)), dtype=torch.uint8, device=device)
        i32 = torch.empty(12, 10, dtype=torch.int32, device=device)
        sc = torch.empty(12, 10, dtype=torch.float32, device=device)
        d64l = torch.empty(12, 10, dtype=torch.float64, device=device)
        st = _lib.stream_ptr(device)
        _lib.check(L.mmr_cosine_topk_ex(qd.data_ptr(), gd.data_ptr(), _lib.MMR_BF16, 12, 9000, 512, 10, 1.0, 0.0, 0,
                                        i32.data_ptr(), sc.data_ptr(), d64l.data_ptr(), 0, ws.data_ptr(), ws.numel(), st))
        packed = torch.empty(12, 10, 2, dtype=torch.int64, device=device)
        _lib.check(L.mmr_topk_pack(i32.data_ptr(), d64l.data_ptr(), 12, 10, offset, packed.data_ptr(), st))
        parts = torch.empty(1, 12, 10, 2, dtype=torch.int64, device=device)
        _lib.check(L.mmr_allgather_topk_packed(comm, packed.data_ptr(), 12, 10, parts.data_ptr(), st))
        torch.cuda.synchronize(device)
        assert torch.equal(parts[0], packed)
        s2, i2, d2 = S.merge_topk_packed(parts, 100.0)
        assert torch.equal(i2, idx) and torch.equal(d2, d64) and torch.equal(s2, score)
        assert L.mmr_allgather_topk_packed(None, packed.data_ptr(), 12, 10, parts.data_ptr(), 0) == -22
        assert L.mmr_allgather_topk_packed(comm, 0, 12, 10, parts.data_ptr(), 0) == -22
    finally:
        L.mmr_comm_destroy(comm)


@pytest.mark.parametrize("E", [128, 512, 768])
def test_fp32_index_with_presplit_gallery_is_identical(S, oracle, device, E):
    """Round 3: a GalleryIndex over an fp32 gallery keeps the gallery's hi / lo bf16 split and its scan
    streams that instead of splitting every tile again (mmr_gallery_split_bf16 + mmr_cosine_topk_split).  Same split function,
    same MFMA operands: indices, fp64 dots, scores AND the per-query path status must equal the per-call path and the oracle --
    ragged last tile, two scan passes (> 128 queries), a crowded boundary that needs the exhaustive path, un-normalised rows."""
    from mmr_amd import search
    N = 50003
    gal = synth.synth_unit_rows(N, E, seed=61) * torch.linspace(0.5, 3.0, N).unsqueeze(1)
    q = synth.synth_unit_rows(131, E, seed=62)
    dup = list(range(700, 40700, 1000))
    v = gal[77] / gal[77].norm() * 3.5                           # the longest row: its 41 copies are query 0's whole top-10
    gal[77] = v
    gal[dup] = v.clone()
    q[0] = v / v.norm()
    gd, qd = gal.to(device), q.to(device)
    plain = search.GalleryIndex(gd, presplit=False)
    split = search.GalleryIndex(gd, presplit=True)
    assert split._split is not None and plain._split is None
    a = plain.search(qd, 10, 100.0, return_dot64=True, return_status=True)
    b = split.search(qd, 10, 100.0, return_dot64=True, return_status=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    oi, os_, od = oracle.cosine_topk(q, gal, 10, scale=100.0)
    assert np.array_equal(b[1].cpu().numpy(), oi) and np.array_equal(b[2].cpu().numpy(), od)
    assert np.array_equal(b[0].cpu().numpy(), os_)
    assert int(b[3][0]) == 1                                      # the crowded query took the exhaustive path in both
    # in-place row updates go through update_rows: bound and split are both re-measured
    rows = torch.tensor([5, 4000, N - 1])
    newv = synth.synth_unit_rows(3, E, seed=63) * 7.0
    split.update_rows(rows, newv)
    gal2 = gal.clone()
    gal2[rows] = newv
    s2, i2, d2 = split.search(qd, 10, 100.0, return_dot64=True)
    oi2, os2, od2 = oracle.cosine_topk(q, gal2, 10, scale=100.0)
    assert np.array_equal(i2.cpu().numpy(), oi2) and np.array_equal(d2.cpu().numpy(), od2)
    # the packed (sharded-search) message uses the same path
    assert torch.equal(split.search_packed(qd[:5], 10, 1.0, 1000)[..., 0] - 1000, i2[:5])
    # bf16 galleries never carry a split
    assert search.GalleryIndex(gd.bfloat16(), presplit=True)._split is None


@pytest.mark.gpu
def test_fp32_split_search_second_tier_certifies_what_the_bf16_tier_cannot(S, oracle, device):
    """The split fp32 index scans the hi array alone first (bf16 operands: the certificate's margin grows by the measured
    rounding residuals, ~3.6e-3 |q||g| here, and 32 candidate tiles are kept) and re-runs only the queries that tier leaves open
    through the three-product scan (margin 8e-5 |q||g|, k + 6 = 16 candidate tiles) before anything reaches the exhaustive
    path.  Query 0 has 40 near neighbours in 40 different tiles whose scores step down by 3e-5: all 40 tiles sit within the
    first tier's margin of the 10th score (not certifiable with 32 candidates), only 3 within the second tier's.  Query 1 has
    41 exact duplicates (no scan can certify it: exhaustive path).  All results equal the oracle and the per-call path."""
    from mmr_amd import search
    N, E = 50003, 512
    gal = synth.synth_unit_rows(N, E, seed=71)
    q = synth.synth_unit_rows(140, E, seed=72)
    u = q[0].double()
    u /= u.norm()
    rows = list(range(500, 40500, 1000))
    w = synth.synth_unit_rows(40, E, seed=73).double()
    w -= (w @ u).unsqueeze(1) * u
    w /= w.norm(dim=1, keepdim=True)
    c = 0.9 - 3e-5 * torch.arange(40, dtype=torch.float64)
    gal[rows] = (c.unsqueeze(1) * u + (1 - c * c).sqrt().unsqueeze(1) * w).float()
    q[0] = u.float()
    dup = list(range(900, 41900, 1000))
    gal[dup] = gal[77].clone()
    q[1] = gal[77].clone()
    gd, qd = gal.to(device), q.to(device)
    tiered = search.GalleryIndex(gd, presplit=True)
    assert tiered._split is not None
    vals, idx, d64, status = tiered.search(qd, 10, 100.0, return_dot64=True, return_status=True)
    oi, os_, od = oracle.cosine_topk(q, gal, 10, scale=100.0)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(d64.cpu().numpy(), od)
    assert np.array_equal(vals.cpu().numpy(), os_)
    assert idx[0].tolist() == rows[:10]
    assert int(status[0]) == 0                                    # certified (by the second tier), not exhaustive
    assert int(status[1]) == 1
    plain = search.GalleryIndex(gd, presplit=False).search(qd, 10, 100.0, return_dot64=True, return_status=True)
    for x, y in zip((vals, idx, d64, status), plain):
        assert torch.equal(x, y)
    # the first tier alone could not have certified query 0: its certificate, recomputed here from its definition
    gh, qh = gal.bfloat16().float(), q[0].bfloat16().float()
    assert abs(tiered._split[2].item() - (gal - gh).norm(dim=1).max().item()) < 1e-6        # measured residual bound
    margin1 = ((q[0] - qh).norm() * 1.0 + q[0].norm() * (1 + 2.0 ** -8) * tiered._split[2].item() + 8e-5).item()
    tile_max = torch.full(((N + 31) // 32,), -1e30)
    tile_max.scatter_reduce_(0, torch.arange(N) // 32, gh @ qh, "amax")
    kth = float(od[0, 9])
    assert 3e-3 < margin1 < 4.2e-3
    assert kth < tile_max.topk(33).values[32].item() + margin1 - 1e-4       # 33rd-best tile still inside the margin
    exact = gal.double() @ q[0].double()
    tile16 = torch.full(((N + 15) // 16,), -1e30, dtype=torch.float64)
    tile16.scatter_reduce_(0, torch.arange(N) // 16, exact, "amax")
    assert kth > tile16.topk(17).values[16].item() + 8e-5 + 5e-5           # second tier: 17th-best tile well outside


@pytest.mark.gpu
@pytest.mark.parametrize("E", [128, 768])
def test_fp32_split_tiers_with_open_queries_in_every_chunk(S, oracle, device, E):
    """The two tiers walk the queries in different chunk sizes (bf16 plan: 256 / 128 queries per pass, fp32 plan: 128 / 64) and
    share one flag array: queries that only the second tier can certify, and ones that need the exhaustive path, are planted
    in the first, a middle and the last chunk of both walks.  Results and path status equal the per-call path and the oracle."""
    from mmr_amd import search
    N, Q = 30011, 300
    gal = synth.synth_unit_rows(N, E, seed=81)
    q = synth.synth_unit_rows(Q, E, seed=82)
    tier2_q, exh_q = [0, 130, 299], [5, 257]
    w = synth.synth_unit_rows(40, E, seed=83).double()
    c = 0.9 - 3e-5 * torch.arange(40, dtype=torch.float64)
    for j, qi in enumerate(tier2_q):                              # 40 near neighbours, 3e-5 apart, in 40 different tiles
        u = q[qi].double()
        u /= u.norm()
        wj = w - (w @ u).unsqueeze(1) * u
        wj /= wj.norm(dim=1, keepdim=True)
        rows = list(range(100 + 37 * j, 100 + 37 * j + 40 * 700, 700))
        gal[rows] = (c.unsqueeze(1) * u + (1 - c * c).sqrt().unsqueeze(1) * wj).float()
        q[qi] = u.float()
    for j, qi in enumerate(exh_q):                                # 41 exact duplicates
        src = 50 + j
        gal[list(range(400 + 53 * j, 400 + 53 * j + 40 * 700, 700))] = gal[src].clone()
        q[qi] = gal[src].clone()
    gd, qd = gal.to(device), q.to(device)
    a = search.GalleryIndex(gd, presplit=True).search(qd, 10, 100.0, return_dot64=True, return_status=True)
    b = search.GalleryIndex(gd, presplit=False).search(qd, 10, 100.0, return_dot64=True, return_status=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    oi, os_, od = oracle.cosine_topk(q, gal, 10, scale=100.0)
    assert np.array_equal(a[1].cpu().numpy(), oi) and np.array_equal(a[2].cpu().numpy(), od)
    assert np.array_equal(a[0].cpu().numpy(), os_)
    st = a[3].cpu()
    assert all(int(st[i]) == 0 for i in tier2_q) and all(int(st[i]) == 1 for i in exh_q)


@pytest.mark.gpu
def test_tiered_fp32_search_is_graph_capturable(S, oracle, device):
    """The tiers decide on the device (flags read by the next tier's kernels), so a GalleryIndex search over a split fp32
    gallery captures into a hipGraph like every other call of include/mmr.h: replayed on new queries -- one that only the second
    tier certifies, one that needs the exhaustive path, ordinary ones -- it returns what the eager call and the oracle return."""
    from mmr_amd import search
    N, E = 20011, 512
    gal = synth.synth_unit_rows(N, E, seed=91)
    qa = synth.synth_unit_rows(9, E, seed=92)
    qb = synth.synth_unit_rows(9, E, seed=93)
    u = qb[3].double()
    u /= u.norm()
    w = synth.synth_unit_rows(40, E, seed=94).double()
    w -= (w @ u).unsqueeze(1) * u
    w /= w.norm(dim=1, keepdim=True)
    c = 0.9 - 3e-5 * torch.arange(40, dtype=torch.float64)
    gal[list(range(60, 60 + 40 * 480, 480))] = (c.unsqueeze(1) * u + (1 - c * c).sqrt().unsqueeze(1) * w).float()
    qb[3] = u.float()
    gal[list(range(200, 200 + 40 * 480, 480))] = gal[7].clone()
    qb[5] = gal[7].clone()
    index = search.GalleryIndex(gal.to(device), presplit=True)
    static_q = qa.to(device).clone()
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        index.search(static_q, 10, 100.0, return_dot64=True, return_status=True)            # warm: workspace allocated
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = index.search(static_q, 10, 100.0, return_dot64=True, return_status=True)
    torch.cuda.current_stream(device).wait_stream(side)
    for q in (qb, qa, qb):
        static_q.copy_(q.to(device))
        graph.replay()
        torch.cuda.synchronize(device)
        oi, os_, od = oracle.cosine_topk(q, gal, 10, scale=100.0)
        assert np.array_equal(out[1].cpu().numpy(), oi) and np.array_equal(out[2].cpu().numpy(), od)
        assert np.array_equal(out[0].cpu().numpy(), os_)
        expect = [0] * 9
        if q is qb:
            expect[5] = 1
        assert out[3].cpu().tolist() == expect


@pytest.mark.gpu
@pytest.mark.slow
def test_full_size_fp32_gallery_tiered_index_exact(S, oracle, device):
    """1M x 512 fp32 (the reference's feature-cache dtype at BASELINE configs[1]'s size), 128 queries, k = 10: the tiered split
    index against the per-call three-product path on ALL queries (idx, fp64 dots, scores), size-independent properties, and the
    CPU oracle on a sample."""
    N, E, Q, k = 1_000_000, 512, 128, 10
    gal = synth.synth_unit_rows(N, E, seed=3)
    q = synth.synth_unit_rows(Q, E, seed=4)
    gd, qd = gal.to(device), q.to(device)
    tiered = S.GalleryIndex(gd, presplit=True)
    vals, idx, d64, status = tiered.search(qd, k, return_dot64=True, return_status=True)
    v2, i2, d2 = S.GalleryIndex(gd, presplit=False).search(qd, k, return_dot64=True)
    assert torch.equal(idx, i2) and torch.equal(d64, d2) and torch.equal(vals, v2)
    idx_c, d_c = idx.cpu(), d64.cpu()
    assert (d_c[:, :-1] >= d_c[:, 1:]).all() and ((idx_c >= 0) & (idx_c < N)).all()
    assert all(len(set(r.tolist())) == k for r in idx_c)
    re = (gal[idx_c.reshape(-1)].double().reshape(Q, k, E) * q.double().unsqueeze(1)).sum(-1)
    assert (re - d_c).abs().max() < 1e-12
    _, i11 = tiered.search(qd[:8], k + 1)
    assert torch.equal(i11[:, :k].cpu(), idx_c[:8])
    sample = [0, 17, Q - 1]
    oi, _, od = oracle.cosine_topk(q[sample], gal, k)
    assert np.array_equal(idx_c[sample].numpy(), oi) and np.array_equal(d_c[sample].numpy(), od)
    print("status (queries on the exhaustive path):", int(status.sum()))
    assert int(status.sum()) == 0
