"""Experiment: the BASELINE step (encode 256 images + top-10 of 256 queries over 1M x 512) with 1, 2 or 3 steps in flight on
as many HIP streams (own model / index workspaces per stream).  Checks that the results equal the one-stream results."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import search, synth

dev = torch.device("cuda:0")
B, L = 256, int(os.environ.get("LANES", 3))
models = []
for _ in range(L):
    m, _p = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
    m.bfloat16()
    models.append(m)
px = torch.randn(B, 3, 224, 224, device=dev).bfloat16()
gal = torch.randn(1_000_000, 512, device=dev)
gal = (gal / gal.norm(dim=-1, keepdim=True)).bfloat16()
q = synth.synth_unit_rows(256, 512, seed=4).bfloat16().to(dev)
idx = [search.GalleryIndex(gal) for _ in range(L)]
streams = [torch.cuda.Stream(dev) for _ in range(L)]
N = int(os.environ.get("N", 60))


def run(lanes):
    cur = torch.cuda.current_stream(dev)
    outs = []
    for s in streams[:lanes]:
        s.wait_stream(cur)
    for i in range(N):
        l = i % lanes
        with torch.cuda.stream(streams[l]):
            f = models[l].encode_image(px, normalize=True)
            o = idx[l].search(q, 10, 1.0)
        if i >= N - lanes:
            outs.append((f, o))
    for s in streams[:lanes]:
        cur.wait_stream(s)
    return outs


ref = None
for lanes in (1, 2, 3, 1, 2, 3)[:2 * L] if L == 3 else (1, 2, 1, 2):
    run(lanes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run(lanes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ref is None:
        ref = outs[0]
    same = all(torch.equal(f, ref[0]) and torch.equal(o[0], ref[1][0]) and torch.equal(o[1], ref[1][1]) for f, o in outs)
    print(f"{lanes} step(s) in flight: {dt / N * 1e3:7.3f} ms per step  {N * B / dt:9.0f} images/s  results equal: {same}", flush=True)
