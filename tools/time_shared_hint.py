"""A/B in one process: two ViT-B/32 forwards in flight with and without the shared-chip tile hint (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import gallery

dev = torch.device("cuda:0")
model, _ = mmr_amd.load(os.environ.get("MODEL", "ViT-B/32"), device=dev, weights="synthetic")
model.bfloat16()
B = int(os.environ.get("B", 256))
S = model.input_resolution
px = torch.randn(B, 3, S, S, device=dev).bfloat16()
N = int(os.environ.get("N", 40))


def run(lanes, hint):
    with gallery._Lanes(dev, lanes, model if hint else None) as L:
        for i in range(N):
            with L.run(i) as lane:
                o = model.encode_image(px, normalize=True, lane=lane)
    return o


ref = None
for lanes, hint in ((2, False), (2, True), (1, False), (1, True), (2, False), (2, True)):
    if lanes == 1 and hint:
        ctx = model.shared_chip()
    else:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        run(lanes, hint)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = run(lanes, hint)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ref is None:
        ref = o.clone()
    print(f"lanes {lanes} hint {int(hint)}: {dt / N * 1e3:7.3f} ms per forward {N * B / dt:9.0f} images/s equal {torch.equal(o, ref)}", flush=True)
