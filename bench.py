#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: batched CLIP ViT-B/32 bf16 encode + cosine top-10 over a
1M x 512 bf16 gallery (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP, per rank (inputs already resident in HBM):
  encode leg : encode_image(256 synthetic 224x224x3 bf16 images) -> 256 L2-normalised bf16 features
  search leg : top-10 of 256 (replicated, seeded) bf16 queries over this rank's 1M x 512 bf16 gallery
               shard; for N > 1 one RCCL all-gather of the packed per-shard top-10 + exact merge.
Weak scaling: per-rank work is fixed; at N ranks a step encodes N*256 images (data-parallel, no
collective) and answers 256 queries over an N x 1M gallery.
`value` = N*256 / step time = images/s through encode+search; the legs are reported separately
(`encode_images_per_s`, `search_queries_per_s`, `search_gpairs_per_s`) from HIP events recorded on
the launch stream inside the same timed region.
`roofline` is for the kernel that dominates the step (the bf16 MFMA GEMM); `roofline_search` for the
HBM-bound gallery scan; their launch durations come from HIP event pairs around every launch in a
second pass over the same K steps (instrumenting the timed region itself would slow it ~15%).  `cpu_baseline` times the CPU oracle / the reference's own torch expression
on the host cores of this box on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODEL = "ViT-B/32"
BATCH = 256
GALLERY_ROWS = 1_000_000
EMBED = 512
TOPK = 10
QUERIES = 256
# dense peaks from /opt/skills/guides/MI355X_MICROARCH.md (spec): bf16 MFMA ~2.5 PFLOP/s, HBM3E 8 TB/s
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def gemm_flops_per_forward(cfg, B):
    """Algorithmic FLOPs of the GEMM launches of one vision forward (2 per multiply-add)."""
    T, d, m, L, E = cfg.tokens, cfg.width, cfg.mlp, cfg.layers, cfg.embed_dim
    rows = B * T
    per_layer = 2 * rows * d * (3 * d) + 2 * rows * d * d + 2 * rows * d * m + 2 * rows * m * d
    patch = 2 * (B * (T - 1)) * cfg.patch_k * d
    proj = 2 * B * d * E
    return patch + L * per_layer + proj, 1 + 4 * L + 1


def host_cores():
    """Threads for the CPU leg: the cores this process may run on, capped at the 16-core share a
    one-GPU box gets on this pool."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


# BASELINE.json's metric string, verbatim
METRIC = "images/sec encode + Mqueries/sec top-10 over 1M\u00d7512 gallery, 1\u21928 MI355X"


def cpu_baseline(model_cfg, cores):
    """Bounded CPU sample on this box's host cores: the fp32 oracle for encode, the reference's torch
    expression for scoring + top-k.  Reported beside the GPU number; never part of it."""
    from mmr_amd import synth, weights
    from oracle import clip_ref, search_ref

    torch.set_num_threads(cores)
    w = weights.make_vision_weights(model_cfg.vision)
    n_img = 2 * BATCH                                         # two steps' images (~9 s of CPU work)
    px = synth.synth_images(n_img, model_cfg.vision.image_size, seed=2)
    with torch.no_grad():
        clip_ref.encode_image(w, model_cfg.vision, px[:4])
        t0 = time.perf_counter()
        for i in range(0, n_img, 128):
            clip_ref.l2_normalize(clip_ref.encode_image(w, model_cfg.vision, px[i:i + 128]))
        t_enc = time.perf_counter() - t0
    enc_ips = n_img / t_enc
    n_gal, n_q, reps = GALLERY_ROWS, QUERIES, 3               # the full gallery, one step's queries, three times (~4 s)
    gal = synth.synth_unit_rows(n_gal, EMBED, seed=3)
    q = synth.synth_unit_rows(n_q, EMBED, seed=4)
    search_ref.reference_expression_topk(gal[:1000], q, TOPK, 100.0)
    t0 = time.perf_counter()
    for _ in range(reps):
        search_ref.reference_expression_topk(gal, q, TOPK, 100.0)
    t_s = time.perf_counter() - t0
    pairs_per_s = reps * n_gal * n_q / t_s
    t_step = BATCH / enc_ips + QUERIES * GALLERY_ROWS / pairs_per_s
    return {
        "value": round(BATCH / t_step, 3), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/clip_ref.py fp32 encode of {n_img} images ({enc_ips:.1f} img/s, {t_enc:.1f} s) + reference "
                   f"torch expression 100*F@q.T + topk({TOPK}) on {n_gal}x{EMBED} fp32 x {n_q} queries x {reps} "
                   f"({pairs_per_s / 1e9:.2f} Gpairs/s, {t_s:.1f} s), scaled to one {BATCH}-image / {QUERIES}-query / "
                   f"{GALLERY_ROWS}-row step; torch {torch.__version__}, {cores} threads"),
        "encode_images_per_s": round(enc_ips, 2), "search_gpairs_per_s": round(pairs_per_s / 1e9, 3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # Native libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on
    # stdout, so everything else is routed to stderr and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path (the CPU oracle is only the baseline leg)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    # under torch.distributed.run (RANK set) always build the process group, also for one rank, so
    # the collective path can be rehearsed on a single GPU; plain `python bench.py` stays group-free
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mmr_amd
    from mmr_amd import _lib, search, synth

    model, _ = mmr_amd.load(MODEL, device=dev)
    model.bfloat16()
    cfg = model.cfg
    g = torch.Generator(device=dev).manual_seed(2 + rank)
    pixels = torch.randn(BATCH, 3, cfg.vision.image_size, cfg.vision.image_size, generator=g, device=dev,
                         dtype=torch.float32).bfloat16()
    g = torch.Generator(device=dev).manual_seed(3 + rank)
    gal = torch.empty(GALLERY_ROWS, EMBED, dtype=torch.bfloat16, device=dev)
    for s in range(0, GALLERY_ROWS, 1 << 18):
        e = min(GALLERY_ROWS, s + (1 << 18))
        x = torch.randn(e - s, EMBED, generator=g, device=dev)
        gal[s:e] = (x / x.norm(dim=-1, keepdim=True)).bfloat16()
    queries = synth.synth_unit_rows(QUERIES, EMBED, seed=4).bfloat16().to(dev)   # replicated on every rank
    index = search.ShardedGalleryIndex(gal, group=None) if use_dist else search.GalleryIndex(gal)

    def step(ev=None):
        if ev:
            ev[0].record()
        feats = model.encode_image(pixels, normalize=True)
        if ev:
            ev[1].record()
        out = index.search(queries, TOPK, 1.0)
        if ev:
            ev[2].record()
        return feats, out

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    fence()
    elapsed = time.perf_counter() - t0

    # Per-kernel durations for the roofline legs: the SAME K steps again with a HIP event pair around
    # every launch, recorded on the launch stream.  Kept out of the timed region on purpose: an event
    # pair per launch (~110 launches/step) serialises the queue and costs ~15% of the step time.
    _lib.prof_enable(True, max(4096, 160 * args.steps))
    for i in range(args.steps):
        step()
    fence()
    _lib.prof_enable(False)
    prof = _lib.prof_read()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    srch_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps

    if rank == 0:
        gflops, n_gemm = gemm_flops_per_forward(cfg.vision, BATCH)
        gemm_ms, gemm_n = prof["gemm"]
        scan_ms, scan_n = prof["scan"]
        fwd = gemm_n / n_gemm if n_gemm else 0
        gemm_tflops = (gflops * fwd) / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        scan_bytes = GALLERY_ROWS * EMBED * 2 + QUERIES * EMBED * 2 + QUERIES * TOPK * 8
        scan_gbs = scan_bytes * scan_n / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))   # newest round's PMC passes
        tf = cands[-1] if cands else ""
        if tf and os.path.exists(tf):
            try:
                traffic = json.load(open(tf))
            except Exception:
                traffic = None
        line = {
            "metric": METRIC,
            "value": round(world * BATCH / (ms_per_step * 1e-3), 1),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{MODEL} bf16 encode batch {BATCH}/GPU + top-{TOPK} of {QUERIES} queries over "
                                   f"{GALLERY_ROWS}x{EMBED} bf16 gallery rows/GPU (BASELINE configs[1])",
                       "encode_batch_per_gpu": BATCH, "gallery_rows_per_gpu": GALLERY_ROWS, "queries": QUERIES,
                       "k": TOPK, "embed_dim": EMBED,
                       "parallelism": f"dp{world} encode, gallery row-sharded x{world}, 1 all-gather of top-k"
                                      if world > 1 else "single GPU",
                       "weights": "seeded random init (no checkpoint reachable offline)"},
            "encode_ms": round(enc_ms, 4), "search_ms": round(srch_ms, 4),
            "encode_images_per_s": round(world * BATCH / (enc_ms * 1e-3), 1),
            "search_queries_per_s": round(QUERIES / (srch_ms * 1e-3), 1),
            "search_mqueries_per_s": round(QUERIES / (srch_ms * 1e-3) / 1e6, 4),
            "search_gpairs_per_s": round(QUERIES * GALLERY_ROWS * world / (srch_ms * 1e-3) / 1e9, 2),
            "roofline": {
                "kernel": "gemm256_bf16_kernel / gemm_bf16_kernel (all epilogues)", "bound": "mfma",
                "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
                "traffic": (traffic or {}).get("gemm_bytes_per_launch"),
                "avg_launch_us": round(gemm_ms / gemm_n * 1e3, 2) if gemm_n else None, "launches": gemm_n,
                "algorithmic_gflop_per_launch": round(gflops / n_gemm / 1e9, 3),
                "measured": "HIP event pairs around each launch on the launch stream, second pass of the same K steps",
            },
            "roofline_search": {
                "kernel": "scan_kernel<512>", "bound": "hbm",
                "achieved": round(scan_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(scan_gbs / PEAK_HBM_GBS, 4),
                "traffic": (traffic or {}).get("scan_bytes_per_launch"),
                "avg_launch_us": round(scan_ms / scan_n * 1e3, 2) if scan_n else None, "launches": scan_n,
                "algorithmic_bytes_per_launch": scan_bytes,
            },
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if k != "dropped"},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, host_cores())
            except Exception as ex:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": host_cores(), "kind": "port",
                                        "sample": f"failed: {ex!r}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
