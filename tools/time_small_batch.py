"""Latency of small-batch encode (the reference's build_cache loop runs batch 1): eager launches vs hipGraph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd

dev = torch.device("cuda:0")
model, _ = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
for B in [int(x) for x in os.environ.get("BS", "1,8,32,64").split(",")]:
    px = torch.randn(B, 3, 224, 224, device=dev)
    for _ in range(3):
        model.encode_image(px)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        f = model.encode_image(px)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e3
    # graph
    g = torch.cuda.CUDAGraph()
    sx = px.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model.encode_image(sx)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model.encode_image(sx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n * 1e3
    ok = torch.equal(out, model.encode_image(sx))
    print(f"B={B:3d}: eager {eager:.3f} ms  graph {graph:.3f} ms  ({B/eager*1e3:.0f} / {B/graph*1e3:.0f} img/s) same={ok}", flush=True)
