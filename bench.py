#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: batched CLIP encode + cosine top-10 over a gallery shard per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5                      # BASELINE.json configs[1] (default)
    python bench.py --config cfg4                                       # one GPU's share of configs[3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--config cfgX]
    python bench.py --gpus N [...]                                      # same thing without a launcher: see below
    python bench.py --gpus 1 --spawn                                    # the N > 1 code path (process group, RCCL
                                                                        # all-gather, pipelined steps) with ONE rank

Launching.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) each process is one rank.  Called
plainly with --gpus N > 1 (or with --spawn), this process becomes a PARENT that never touches the GPU: it picks a free
port on 127.0.0.1, starts N fresh children of this same file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (plain
subprocesses -- no exec of a process that has initialised the GPU), relays rank 0's single JSON line, and exits non-zero
if any child does (the others are then stopped by PID).  A rank whose GPU index does not exist on the node exits with a
device-count message, so `--gpus 2` on a one-GPU box fails clearly instead of hanging in the rendezvous.

Workloads (--config; names follow SURVEY.md section 8d; per RANK, weak scaling -- fixed per-GPU work):
  cfg2  BASELINE configs[1]: ViT-B/32 bf16 encode of 256 images + top-10 of 256 queries over 1M x 512 bf16 rows
  cfg3  BASELINE configs[2]: CLIP text tower bf16 on 256 prompts, their features are the queries over 1M x 512 rows
  cfg4  BASELINE configs[3]: search only, 1.25M x 512 rows per rank (8 ranks = the 10M gallery), 256 queries
  cfg5  BASELINE configs[4]: ViT-L/14@336 bf16 encode of 128 images per rank + top-10 of 128 queries over 1M x 768 rows
  build the reference's gallery loop from RAW images (code/search_image.py:142-165): 256 uint8 480x640 images per step ->
        Pillow-exact preprocess -> ViT-B/32 encode -> normalise -> gallery rows, preprocess overlapped with the previous
        batch's encode; reported beside the serial schedule and encode-only (no search leg)
One STEP, per rank (inputs already resident in HBM): the config's encode leg (none for cfg4), then the search leg:
local top-10 over this rank's shard; for N > 1 ONE RCCL all-gather (all_gather_into_tensor, async) of the packed
per-shard top-10 and an exact merge.  Steps are independent batches, and --lanes of them (default 2) are in flight at
once: step i runs on HIP stream i % lanes with the model / index workspace of that lane (`encode_image(lane=)`,
`search(lane=)`), so one step's 150-200-tile GEMM launches and bandwidth-bound row kernels share the chip with the
other's (+8-10 % on every config; bit-identical results).  With N > 1 a lane's all-gather is in flight while that lane's
next step encodes and scans.  All K steps, merges included, complete inside the timed region.
`value` = units of the config (images, texts or queries) all ranks processed per second through the whole step.
The legs (`encode_ms`, `search_ms`) and `one_step_in_flight` come from a second pass of the same K steps strictly one after
another (HIP events on the launch stream): with two steps in flight a leg's events would span the other lane's kernels.
`roofline` is for the kernel that dominates the step (the bf16 MFMA GEMM; the gallery scan for cfg4);
`roofline_search` for the gallery scan, which at 256 resident queries is paced by HBM and MFMA alike -- both
fractions are given.  Launch durations come from HIP event pairs around every launch in a second pass over the same
K steps (instrumenting the timed region itself would slow it ~15%).  `search_sweep` (cfg2) times the search at
Q = 1, 32, 256, 1024 queries.  After the timed region the last step's result is VERIFIED
(`verify`): every returned global id is re-scored by the rank that owns it (fp64) and must equal the merged dot,
the list must be sorted / unique, and the merged k-th dot must be >= every rank's local (k+1)-th -- together that
is the proof that the sharded result is the exact global top-k.  `cpu_baseline` times the CPU oracle / the
reference's own torch expression on the host cores of this box on a bounded sample (rank 0, N = 1, cfg2 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = None  # imported by main() AFTER the launch decision: the parent of self-spawned ranks never needs it

TOPK = 10
# dense peaks from /opt/skills/guides/MI355X_MICROARCH.md (spec): bf16 MFMA ~2.5 PFLOP/s, HBM3E 8 TB/s
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    # name: (model, encode kind, batch per rank, gallery rows per rank, queries, BASELINE.json entry)
    "cfg2": dict(model="ViT-B/32", encode="image", batch=256, rows=1_000_000, queries=256, baseline="configs[1]"),
    "cfg3": dict(model="ViT-B/32", encode="text", batch=256, rows=1_000_000, queries=256, baseline="configs[2]"),
    "cfg4": dict(model="ViT-B/32", encode="none", batch=0, rows=1_250_000, queries=256, baseline="configs[3]"),
    "cfg5": dict(model="ViT-L/14@336px", encode="image", batch=128, rows=1_000_000, queries=128, baseline="configs[4]"),
    # the reference's gallery loop end to end (code/search_image.py:142-165): raw uint8 VGA images -> preprocess -> encode ->
    # normalise -> rows of a preallocated gallery; preprocess of batch i+1 overlapped with the encode of batch i
    "build": dict(model="ViT-B/32", encode="build", batch=256, rows=0, queries=0, baseline="configs[1] encode leg, from raw images"),
}


def gemm_flops_per_forward(cfg, B):
    """Algorithmic FLOPs of the GEMM launches of one tower forward (2 per multiply-add) and their count."""
    T, d, m, L, E = cfg.tokens, cfg.width, cfg.mlp, cfg.layers, cfg.embed_dim
    rows = B * T
    per_layer = 2 * rows * d * (3 * d) + 2 * rows * d * d + 2 * rows * d * m + 2 * rows * m * d
    proj = 2 * B * d * E
    if cfg.kind == "vision":
        patch = 2 * (B * (T - 1)) * cfg.patch_k * d
        return patch + L * per_layer + proj, 1 + 4 * L + 1
    return L * per_layer + proj, 4 * L + 1


def host_cores():
    """Threads for the CPU leg: the cores this process may run on, capped at the 16-core share a
    one-GPU box gets on this pool."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


# BASELINE.json's metric string, verbatim
METRIC = "images/sec encode + Mqueries/sec top-10 over 1M×512 gallery, 1→8 MI355X"


def cpu_baseline(model_cfg, cores, batch, rows, queries, embed):
    """Bounded CPU sample on this box's host cores: the fp32 oracle for encode, the reference's torch
    expression for scoring + top-k.  Reported beside the GPU number; never part of it."""
    from mmr_amd import synth, weights
    from oracle import clip_ref, search_ref

    torch.set_num_threads(cores)
    w = weights.make_vision_weights(model_cfg.vision)
    n_img = 2 * batch                                         # two steps' images (~9 s of CPU work)
    px = synth.synth_images(n_img, model_cfg.vision.image_size, seed=2)
    with torch.no_grad():
        clip_ref.encode_image(w, model_cfg.vision, px[:4])
        t0 = time.perf_counter()
        for i in range(0, n_img, 128):
            clip_ref.l2_normalize(clip_ref.encode_image(w, model_cfg.vision, px[i:i + 128]))
        t_enc = time.perf_counter() - t0
    enc_ips = n_img / t_enc
    reps = 3                                                  # the full gallery, one step's queries, three times (~4 s)
    gal = synth.synth_unit_rows(rows, embed, seed=3)
    q = synth.synth_unit_rows(queries, embed, seed=4)
    search_ref.reference_expression_topk(gal[:1000], q, TOPK, 100.0)
    t0 = time.perf_counter()
    for _ in range(reps):
        search_ref.reference_expression_topk(gal, q, TOPK, 100.0)
    t_s = time.perf_counter() - t0
    pairs_per_s = reps * rows * queries / t_s
    t_step = batch / enc_ips + queries * rows / pairs_per_s
    return {
        "value": round(batch / t_step, 3), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/clip_ref.py fp32 encode of {n_img} images ({enc_ips:.1f} img/s, {t_enc:.1f} s) + reference "
                   f"torch expression 100*F@q.T + topk({TOPK}) on {rows}x{embed} fp32 x {queries} queries x {reps} "
                   f"({pairs_per_s / 1e9:.2f} Gpairs/s, {t_s:.1f} s), scaled to one {batch}-image / {queries}-query / "
                   f"{rows}-row step; torch {torch.__version__}, {cores} threads"),
        "encode_images_per_s": round(enc_ips, 2), "search_gpairs_per_s": round(pairs_per_s / 1e9, 3),
    }


def unit_rows_on_device(n, e, seed, dev):
    """[n,e] L2-normalised bf16 rows generated on the device (a 1-10M row shard is too slow to make on the host)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty(n, e, dtype=torch.bfloat16, device=dev)
    for s in range(0, n, 1 << 18):
        x = torch.randn(min(n, s + (1 << 18)) - s, e, generator=g, device=dev)
        out[s:s + x.shape[0]] = (x / x.norm(dim=-1, keepdim=True)).bfloat16()
    return out


def spawn_ranks(n, argv):
    """Parent of self-started ranks (module docstring, "Launching").  Never imports torch, never touches the GPU."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child_argv = [a for a in argv if a != "--spawn"]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MMR_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs across processes on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + child_argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    rc, out0 = 0, b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                if r == 0:
                    try:
                        o, _ = procs[0].communicate(timeout=0.2)     # drains the pipe while waiting
                        out0 += o
                    except subprocess.TimeoutExpired:
                        continue
                else:
                    try:
                        procs[r].wait(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        continue
                pending.discard(r)
                if procs[r].returncode != 0 and rc == 0:
                    rc = procs[r].returncode or 1
                    print(f"bench.py: rank {r} exited with code {procs[r].returncode}; stopping the other ranks", file=sys.stderr)
                    for j in pending:                               # exact PIDs we started, nothing by pattern
                        procs[j].terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    raise SystemExit(rc if rc >= 0 else 1)


def build_leg(args, C, custom, dev, rank, world, use_dist, dist):
    """--config build: K steps of  uint8 [B,480,640,3] (resident) -> Pillow-exact preprocess -> ViT encode -> L2-normalise ->
    rows of a preallocated gallery (gallery.build_gallery_overlapped), in three schedules: two batches in flight on two
    streams / model workspaces (the default), everything back to back on one stream, and the preprocess of batch i+1 on a
    side stream under the encode of batch i.  Timed beside them in the same process: encode alone on already-preprocessed
    pixels with one and with two batches in flight, and the preprocess alone.  Every schedule's gallery must equal the serial
    one bit for bit, and the serial one the plain per-batch preprocess_batch + encode_image."""
    import mmr_amd
    from mmr_amd import _lib, gallery, preprocess

    B, K, Wm = C["batch"], args.steps, args.warmup
    H, W = 480, 640
    model, _ = mmr_amd.load(C["model"], device=dev, weights="synthetic")
    model.bfloat16()
    E, S = model.cfg.embed_dim, model.input_resolution
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    raws = [torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]

    def batches(n):
        return (raws[i & 1] for i in range(n))

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn):
        fence()
        t0 = time.perf_counter()
        r = fn()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), r

    gal = torch.empty(K * B, E, dtype=torch.bfloat16, device=dev)
    gallery.build_gallery_overlapped(model, batches(Wm + 2), total=(Wm + 2) * B, overlap=True)   # warm-up (both slots)
    t_ovl, g_ovl = timed(lambda: gallery.build_gallery_overlapped(model, batches(K), gallery=gal, overlap=True))
    g_ovl = g_ovl.clone()
    gallery.build_gallery_overlapped(model, batches(2), total=2 * B, lanes=1)
    t_ser, g_ser = timed(lambda: gallery.build_gallery_overlapped(model, batches(K), gallery=gal, lanes=1))
    g_ser = g_ser.clone()
    gallery.build_gallery_overlapped(model, batches(4), total=4 * B, lanes=2)
    t_two, g_two = timed(lambda: gallery.build_gallery_overlapped(model, batches(K), gallery=gal, lanes=2))
    g_two = g_two.clone()
    px = [preprocess.preprocess_batch(list(r), S, out_dtype=torch.bfloat16) for r in raws]

    def enc_only(lanes):
        def run():
            with gallery._Lanes(dev, lanes) as L:
                for i in range(K):
                    with L.run(i) as lane:
                        model.encode_image(px[i & 1], normalize=True, out=gal[i * B:(i + 1) * B], lane=lane)
            return gal
        return run

    enc_only(1)()
    t_enc, g_enc = timed(enc_only(1))
    g_enc = g_enc.clone()
    enc_only(2)()
    t_enc2, g_enc2 = timed(enc_only(2))
    pre = preprocess.UniformBatchPreprocessor(B, H, W, S, device=dev, slots=1)

    def pre_only():
        for i in range(K):
            pre(raws[i & 1], 0)

    pre_only()
    t_pre, _ = timed(pre_only)
    same_ovl = bool(torch.equal(g_ovl, g_ser)) and bool(torch.equal(g_two, g_ser))
    same_ref = bool(torch.equal(g_ser, g_enc)) and bool(torch.equal(g_enc2, g_enc))
    ok = same_ovl and same_ref and bool(torch.isfinite(g_ovl.float()).all())
    gflops, n_gemm = gemm_flops_per_forward(model.cfg.vision, B)
    # roofline of the dominant kernel (the bf16 GEMMs), launch durations from a second, instrumented pass
    _lib.prof_enable(True, max(4096, 300 * K))
    gallery.build_gallery_overlapped(model, batches(K), gallery=gal, lanes=1)
    torch.cuda.synchronize(dev)
    _lib.prof_enable(False)
    prof = _lib.prof_read()
    gemm_ms, gemm_n = prof["gemm"]
    gemm_tflops = gflops * (gemm_n / n_gemm) / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    imgs = world * K * B
    t_best = min(t_ser, t_ovl, t_two)
    line = {
        "metric": METRIC, "value": round(imgs / t_best, 1), "unit": "images/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(t_best / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": (f"gallery build from raw images: {B} uint8 {H}x{W}x3 images/GPU per step (resident) -> bicubic "
                                f"resize + centre crop + normalise ({S}x{S}, Pillow-exact) -> {C['model']} bf16 encode -> "
                                f"L2-normalise -> rows of a preallocated [{K * B},{E}] bf16 gallery (BASELINE {C['baseline']}"
                                f"{'; CUSTOM ' + ' '.join(custom) if custom else ''})"),
                   "name": "build", "encode_batch_per_gpu": B, "image_hw": [H, W],
                   "parallelism": f"dp{world} (no collective)" if use_dist else "single GPU",
                   "schedule": ("value = the fastest of the schedules measured: two batches in flight, each preprocess -> encode on its own "
                                "stream and model workspace (the library default) / back to back on one stream / preprocess of "
                                "batch i+1 on a low-priority side stream under the encode of batch i"),
                   "weights": "seeded random init (no checkpoint reachable offline)"},
        "verify": "ok" if ok else "FAILED",
        "verify_detail": {"every_schedule_equals_serial_bitwise": same_ovl, "serial_equals_preprocess_batch_then_encode_bitwise": same_ref},
        "build_two_lanes_images_per_s": round(imgs / t_two, 1),
        "build_overlapped_images_per_s": round(imgs / t_ovl, 1), "build_serial_images_per_s": round(imgs / t_ser, 1),
        "encode_only_images_per_s": round(imgs / t_enc, 1), "encode_only_two_lanes_images_per_s": round(imgs / t_enc2, 1),
        "preprocess_only_images_per_s": round(imgs / t_pre, 1),
        "two_lanes_over_encode_only": round(t_enc / t_two, 4), "two_lanes_over_encode_only_two_lanes": round(t_enc2 / t_two, 4),
        "overlapped_over_encode_only": round(t_enc / t_ovl, 4), "serial_over_encode_only": round(t_enc / t_ser, 4),
        "roofline": {"kernel": "bf16 MFMA GEMMs of the image tower (all epilogues)", "bound": "mfma", "achieved": round(gemm_tflops, 2),
                     "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4), "traffic": None,
                     "avg_launch_us": round(gemm_ms / gemm_n * 1e3, 2) if gemm_n else None, "launches": gemm_n,
                     "measured": "HIP event pairs around each launch, second pass (serial schedule)"},
    }
    return line, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="workload (SURVEY 8d names); default cfg2 = BASELINE configs[1]")
    ap.add_argument("--model", default=None, help="override the config's model (custom workload)")
    ap.add_argument("--batch", type=int, default=None, help="override images/texts per rank per step")
    ap.add_argument("--rows", type=int, default=None, help="override gallery rows per rank")
    ap.add_argument("--queries", type=int, default=None, help="override queries per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--lanes", type=int, default=None,
                    help="steps in flight (HIP streams with their own model / index workspaces); 1 = strictly one after another; "
                         "default 2 for the configs with an encode leg, 1 for search-only cfg4 (two gallery scans at once are "
                         "6 %% slower than one after the other)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks from this process (implied by --gpus N > 1 without a launcher); with --gpus 1 it "
                         "rehearses the process-group / all-gather path on one GPU")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        spawn_ranks(args.gpus, sys.argv[1:])            # does not return

    global torch
    import torch

    # Native libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on
    # stdout, so everything else is routed to stderr and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    C = dict(CONFIGS[args.config])
    custom = []
    for key in ("model", "batch", "rows", "queries"):
        v = getattr(args, key)
        if v is not None and v != C[key]:
            C[key] = v
            custom.append(f"{key}={v}")
    MODEL, ENC, BATCH, ROWS, QUERIES = C["model"], C["encode"], C["batch"], C["rows"], C["queries"]
    if ENC == "text" and QUERIES != BATCH:
        QUERIES = BATCH                      # the encoded prompts ARE the queries

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE={world})")
    ndev = torch.cuda.device_count()                     # does not initialise the GPU
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X; there is no CPU path (the CPU oracle is only the baseline leg)")
    if local_rank >= ndev:
        raise SystemExit(f"bench.py: rank {rank} of {world} needs GPU index {local_rank}, but this node exposes {ndev} device(s): "
                         f"--gpus {args.gpus} wants one MI355X per rank (RCCL does not run two ranks on one device)")
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

    if ENC == "build":
        line, ok = build_leg(args, C, custom, dev, rank, world, use_dist, dist)
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if not ok:
            raise SystemExit("bench build leg: verification failed")
        return

    model, _ = mmr_amd.load(MODEL, device=dev, weights="synthetic")     # no checkpoint is reachable offline
    model.bfloat16()
    cfg = model.cfg
    EMBED = cfg.embed_dim
    tower_cfg = cfg.text if ENC == "text" else cfg.vision
    pixels = ids = None
    if ENC == "image":
        g = torch.Generator(device=dev).manual_seed(2 + rank)
        pixels = torch.randn(BATCH, 3, cfg.vision.image_size, cfg.vision.image_size, generator=g, device=dev,
                             dtype=torch.float32).bfloat16()
    elif ENC == "text":
        ids = synth.synth_token_ids(BATCH, cfg.text.tokens, cfg.text.vocab, seed=5).to(dev)   # replicated: same queries on every rank
    gal = unit_rows_on_device(ROWS, EMBED, 3 + rank, dev)
    queries = synth.synth_unit_rows(QUERIES, EMBED, seed=4).bfloat16().to(dev)   # replicated on every rank
    index = search.ShardedGalleryIndex(gal, group=None) if use_dist else search.GalleryIndex(gal)

    LANES = max(1, args.lanes) if args.lanes is not None else (1 if ENC == "none" else 2)
    main_stream = torch.cuda.current_stream(dev)
    lane_streams = [torch.cuda.Stream(dev) for _ in range(LANES)] if LANES > 1 else [main_stream]

    def encode(lane=0):
        if ENC == "image":
            return model.encode_image(pixels, normalize=True, lane=lane)
        if ENC == "text":
            return model.encode_text(ids, normalize=True, lane=lane)
        return None

    def run_steps(n, events=None, lanes=LANES):
        """n steps, `lanes` of them in flight: step i runs on HIP stream i % lanes with the model / index workspace of the same
        number (independent batches: the second fills the CUs the first one's 150-200-tile GEMM launches and row kernels
        leave idle).  With a process group the all-gather of a lane's step overlaps that lane's next step (drained at the
        end).  Every step's work, merges included, is enqueued before this returns; the caller's fence waits for it."""
        streams = lane_streams[:lanes] if lanes > 1 else [main_stream]
        if lanes > 1:
            for st in streams:
                st.wait_stream(main_stream)
        pending = [None] * lanes
        def step(i):
            lane = i % lanes
            with torch.cuda.stream(streams[lane]):
                ev = events[i] if events else None
                if ev:
                    ev[0].record()
                feats = encode(lane)
                if ev:
                    ev[1].record()
                q = feats if ENC == "text" else queries
                if use_dist:
                    nxt = index.search_async(q, TOPK, 1.0, lane=lane)
                    if pending[lane] is not None:
                        pending[lane].result()
                    pending[lane] = nxt
                else:
                    index.search(q, TOPK, 1.0, lane=lane)
                if ev:
                    ev[2].record()

        with model.shared_chip(lanes > 1):           # tile policy for forwards that share the chip (mmr_tower_set_shared_chip)
            for i in range(n):
                step(i)
        for lane in range(lanes):
            if pending[lane] is not None:
                with torch.cuda.stream(streams[lane]):
                    pending[lane].result()
        if lanes > 1:
            for st in streams:
                main_stream.wait_stream(st)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    import gc
    run_steps(args.warmup)
    fence()
    gc.collect()
    gc.disable()                 # a collection pause in the launching thread would be charged to a 10-60 ms timed region
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()

    # The two legs, un-overlapped: the same K steps with ONE step in flight, HIP events around encode and search on the
    # launch stream (with several steps in flight a leg's events also span the other lanes' kernels).
    run_steps(min(args.warmup, 2), lanes=1)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    fence()
    t1 = time.perf_counter()
    run_steps(args.steps, events, lanes=1)
    fence()
    elapsed_one = time.perf_counter() - t1

    # Per-kernel durations for the roofline legs: the SAME K steps again with a HIP event pair around
    # every launch, recorded on the launch stream.  Kept out of the timed region on purpose: an event
    # pair per launch serialises the queue and costs ~15% of the step time.
    _lib.prof_enable(True, max(4096, 700 * args.steps))
    run_steps(args.steps, lanes=1)
    fence()
    _lib.prof_enable(False)
    prof = _lib.prof_read()

    t = torch.tensor([elapsed, elapsed_one], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, elapsed_one = float(t[0].item()), float(t[1].item())
    ms_per_step = elapsed / args.steps * 1e3
    ms_per_step_one = elapsed_one / args.steps * 1e3
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in events) / args.steps
    srch_ms = sum(e[1].elapsed_time(e[2]) for e in events) / args.steps

    # ---- outside the timed region: verification, fallback count, query sweep
    vq = encode() if ENC == "text" else queries
    verify, verify_detail = search.verify_exact_topk(index, gal, vq, TOPK)
    local_index = index._index if use_dist else index
    status = local_index.search(vq, TOPK, 1.0, return_status=True)[-1]
    fallback = int(status.sum())

    def scan_bytes_for(q):
        return ROWS * EMBED * 2 + q * EMBED * 2 + q * TOPK * 8     # gallery once + queries + results

    sweep = None
    if args.config == "cfg2" and not args.no_sweep:
        sweep = {}
        for Q in (1, 32, 256, 1024):
            qs = synth.synth_unit_rows(Q, EMBED, seed=40 + Q).bfloat16().to(dev)
            for _ in range(3):
                local_index.search(qs, TOPK, 1.0)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                local_index.search(qs, TOPK, 1.0)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            _lib.prof_enable(True, 4096)
            for _ in range(reps):
                local_index.search(qs, TOPK, 1.0)
            torch.cuda.synchronize(dev)
            _lib.prof_enable(False)
            p = _lib.prof_read()
            scan_ms_q = p["scan"][0] / reps                      # all scan passes of one search (4 passes at Q=1024)
            passes = max(1, p["scan"][1] // reps)
            gbs = (ROWS * EMBED * 2 * passes + Q * EMBED * 2 + Q * TOPK * 8) / (scan_ms_q * 1e-3) / 1e9
            sweep[str(Q)] = {"search_ms": round(ms, 4), "queries_per_s": round(Q / (ms * 1e-3), 1),
                             "gpairs_per_s": round(Q * ROWS / (ms * 1e-3) / 1e9, 2),
                             "scan_ms": round(scan_ms_q, 4), "scan_passes": passes, "scan_gb_per_s": round(gbs, 1),
                             "scan_frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4),
                             "scan_tflops": round(2.0 * Q * ROWS * EMBED / (scan_ms_q * 1e-3) / 1e12, 1)}

    if rank == 0:
        gemm_ms, gemm_n = prof["gemm"]
        scan_ms, scan_n = prof["scan"]
        gemm_tflops, gflops, n_gemm = 0.0, 0, 0
        if ENC != "none":
            gflops, n_gemm = gemm_flops_per_forward(tower_cfg, BATCH)
            fwd = gemm_n / n_gemm if n_gemm else 0
            gemm_tflops = (gflops * fwd) / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        scan_bytes = scan_bytes_for(QUERIES)
        scan_gbs = scan_bytes * scan_n / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        scan_tflops = 2.0 * QUERIES * ROWS * EMBED * scan_n / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
        traffic, tf = None, ""
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))   # newest round's PMC passes
        if cands and args.config == "cfg2" and not custom:
            tf = cands[-1]
            try:
                traffic = json.load(open(tf))
            except Exception:
                traffic = None
        traffic_src = (f"{os.path.relpath(tf, ROOT)} (replayed from the committed rocprofv3 --pmc passes of this workload; "
                       "NOT measured in this run)") if traffic else None
        unit = {"image": "images/s", "text": "texts/s", "none": "queries/s"}[ENC]
        per_step_units = world * BATCH if ENC != "none" else QUERIES
        enc_name = {"image": f"{MODEL} bf16 encode batch {BATCH}/GPU + ", "text": f"{MODEL} text tower bf16 on {BATCH} prompts (= the queries) + ",
                    "none": ""}[ENC]
        workload = (f"{enc_name}top-{TOPK} of {QUERIES} queries over {ROWS}x{EMBED} bf16 gallery rows/GPU "
                    f"(BASELINE {C['baseline']}{'; CUSTOM ' + ' '.join(custom) if custom else ''})")
        roof_gemm = {
            "kernel": "gemm256_persist_kernel / gemm256_bf16_kernel / gemm_bf16_kernel / gemm_skinny_kernel (all epilogues)", "bound": "mfma",
            "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4),
            "traffic": (traffic or {}).get("gemm_bytes_per_launch"), "traffic_source": traffic_src,
            "avg_launch_us": round(gemm_ms / gemm_n * 1e3, 2) if gemm_n else None, "launches": gemm_n,
            "algorithmic_gflop_per_launch": round(gflops / n_gemm / 1e9, 3) if n_gemm else None,
            "measured": "HIP event pairs around each launch on the launch stream, a separate pass of the same K steps with ONE "
                        "step in flight (event pairs inflate every launch: `frac` is a floor; with two steps in flight a "
                        "launch's duration would include the time it shares the chip with the other lane's kernels) -- the "
                        "rocprofv3 summary that matches it is the `--lanes 1` one",
            # the same algorithmic GEMM FLOPs over the UN-instrumented encode time of the one-step-in-flight pass (which also
            # holds the LayerNorm / attention / row kernels): the end-to-end figure; the kernel-level truth lies between the
            # two and is what profiles/*_kernel_stats.csv (rocprofv3) gives
            "uninstrumented_encode_tflops": round(gflops / (enc_ms * 1e-3) / 1e12, 2) if enc_ms > 0 else None,
            "uninstrumented_encode_frac": round(gflops / (enc_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if enc_ms > 0 else None,
            # and over the WHOLE timed step with `steps_in_flight` steps sharing the chip (search time included, its FLOPs not):
            # the figure the +10 % of the second lane shows up in
            "timed_step_tflops": round(gflops / (ms_per_step * 1e-3) / 1e12, 2),
            "timed_step_frac": round(gflops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
        }
        roof_scan = {
            "kernel": f"scan_kernel<{EMBED}>", "bound": "hbm+mfma",
            "achieved": round(scan_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(scan_gbs / PEAK_HBM_GBS, 4),
            "mfma_achieved_tflops": round(scan_tflops, 1), "mfma_frac": round(scan_tflops / PEAK_BF16_TFLOPS, 4),
            "note": f"at {QUERIES} resident queries the scan does {QUERIES} FLOP per gallery byte: it is paced by the matrix pipe "
                    "as much as by HBM (both fractions given); at Q <= 64 it is HBM-paced (see search_sweep)",
            "traffic": (traffic or {}).get("scan_bytes_per_launch"), "traffic_source": traffic_src,
            "avg_launch_us": round(scan_ms / scan_n * 1e3, 2) if scan_n else None, "launches": scan_n,
            "algorithmic_bytes_per_launch": scan_bytes,
        }
        line = {
            "metric": METRIC,
            "value": round(per_step_units / (ms_per_step * 1e-3), 1),
            "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "name": args.config,
                       "encode_batch_per_gpu": BATCH, "gallery_rows_per_gpu": ROWS, "gallery_rows_total": ROWS * world,
                       "queries": QUERIES, "k": TOPK, "embed_dim": EMBED,
                       "parallelism": (f"dp{world} encode (no collective), gallery row-sharded x{world}, 1 async all-gather of "
                                       f"top-k per step overlapped with the next step") if use_dist else "single GPU",
                       "launch": ("self-spawned ranks" if os.environ.get("MMR_BENCH_SPAWNED") else
                                  "torch.distributed.run" if use_dist else "single process"),
                       "steps_in_flight": LANES,
                       "weights": "seeded random init (no checkpoint reachable offline)"},
            "verify": verify, "verify_detail": verify_detail,
            # the same K steps strictly one after another (second pass, same process): what `--lanes 1` measures; the legs
            # below (encode_ms, search_ms and the rates derived from them) are from THAT pass
            "one_step_in_flight": {"ms_per_step": round(ms_per_step_one, 4),
                                   "value": round(per_step_units / (ms_per_step_one * 1e-3), 1)},
            "search_fallback_queries": fallback,
            "encode_ms": round(enc_ms, 4), "search_ms": round(srch_ms, 4),
            "search_queries_per_s": round(QUERIES / (srch_ms * 1e-3), 1),
            "search_mqueries_per_s": round(QUERIES / (srch_ms * 1e-3) / 1e6, 4),
            "search_gpairs_per_s": round(QUERIES * ROWS * world / (srch_ms * 1e-3) / 1e9, 2),
            "roofline": roof_gemm if ENC != "none" else roof_scan,
            "roofline_search": roof_scan,
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if k != "dropped"},
        }
        if ENC != "none":
            key = "encode_images_per_s" if ENC == "image" else "encode_texts_per_s"
            line[key] = round(world * BATCH / (enc_ms * 1e-3), 1)
        if sweep is not None:
            line["search_sweep"] = sweep
        if world == 1 and args.config == "cfg2" and not custom and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, host_cores(), BATCH, ROWS, QUERIES, EMBED)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": host_cores(), "kind": "port",
                                        "sample": f"failed: {ex!r}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not verify.startswith("ok"):
        raise SystemExit(f"bench verification failed: {verify}")


if __name__ == "__main__":
    main()
