"""Experiment: small-batch forwards (the reference's batch-1 and batch-10 loops) with L of them in flight on L lanes / HIP
streams: a batch-1 forward is ~100 dependent launches that leave the chip almost idle, so independent requests overlap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import gallery

dev = torch.device("cuda:0")
model, _ = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
model.bfloat16()
N = int(os.environ.get("N", 400))
for B in (1, 10, 32):
    px = [torch.randn(B, 3, 224, 224, device=dev).bfloat16() for _ in range(8)]
    ref = [model.encode_image(p, normalize=True).clone() for p in px]
    for lanes in (1, 2, 4, 8):
        def run():
            outs = []
            with gallery._Lanes(dev, lanes) as L:
                for i in range(N):
                    with L.run(i) as lane:
                        o = model.encode_image(px[i % 8], normalize=True, lane=lane)
                    if i >= N - 8:
                        outs.append((i % 8, o))
            return outs
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(o, ref[k]) for k, o in outs)
        print(f"batch {B:2d}, {lanes} in flight: {dt / N * 1e3:6.3f} ms per forward  {N * B / dt:8.0f} images/s  equal: {same}", flush=True)
