"""Experiment: N images through the image tower as (a) one forward of batch N, (b) two concurrent forwards of N/2 on two lanes,
for N = 256, 512, 1024 (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import gallery

dev = torch.device("cuda:0")
model, _ = mmr_amd.load(os.environ.get("MODEL", "ViT-B/32"), device=dev, weights="synthetic")
model.bfloat16()
model.max_batch = 4096
S = model.input_resolution
REP = int(os.environ.get("REP", 20))


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REP * 1e3


for N in [int(x) for x in os.environ.get("NS", "256,512,1024").split(",")]:
    px = torch.randn(N, 3, S, S, device=dev).bfloat16()
    one = lambda: model.encode_image(px, normalize=True)

    def two():
        with gallery._Lanes(dev, 2) as L:
            for i in range(2):
                with L.run(i) as lane:
                    model.encode_image(px[i * N // 2:(i + 1) * N // 2], normalize=True, lane=lane)

    def four():
        with gallery._Lanes(dev, 2) as L:
            for i in range(4):
                with L.run(i) as lane:
                    model.encode_image(px[i * N // 4:(i + 1) * N // 4], normalize=True, lane=lane)
    a, b, c = timeit(one), timeit(two), timeit(four)
    print(f"N={N:5d}: one forward {a:7.3f} ms ({N / a:6.1f} img/ms) | 2 x N/2 on two lanes {b:7.3f} ms ({N / b:6.1f}) | 4 x N/4 on two lanes {c:7.3f} ms ({N / c:6.1f})", flush=True)
