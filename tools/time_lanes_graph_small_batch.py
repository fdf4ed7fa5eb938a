"""Experiment: batch-1 / batch-10 forwards captured as hipGraphs, one per lane, replayed with L in flight.  Eager launching
is bound by the host (~110 launches x 3 us per forward); a graph replay costs the host one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd

dev = torch.device("cuda:0")
model, _ = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
model.bfloat16()
N = int(os.environ.get("N", 800))
for B in (1, 10):
    src = torch.randn(B, 3, 224, 224, device=dev).bfloat16()
    ref = model.encode_image(src, normalize=True).clone()
    for lanes in (1, 2, 4, 8, 16):
        streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
        graphs, outs, ins = [], [], []
        for l in range(lanes):
            x = src.clone()
            with torch.cuda.stream(streams[l]):
                model.encode_image(x, normalize=True, lane=l)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[l]):
                    o = model.encode_image(x, normalize=True, lane=l)
            graphs.append(g); outs.append(o); ins.append(x)
        torch.cuda.synchronize()

        def run():
            for i in range(N):
                l = i % lanes
                with torch.cuda.stream(streams[l]):
                    graphs[l].replay()
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(o, ref) for o in outs)
        print(f"batch {B:2d}, {lanes:2d} graphs in flight: {dt / N * 1e3:6.3f} ms per forward  {N * B / dt:8.0f} images/s  equal: {same}", flush=True)
