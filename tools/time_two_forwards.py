"""Experiment: two full-batch forwards in flight on two HIP streams (two model instances, own workspaces) against the same
forwards back to back on one stream -- do the 150-200-tile GEMM launches and the bandwidth-bound row kernels of one forward
fill the CUs the other leaves idle?  (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 256))
models = []
for _ in range(2):
    m, _p = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
    m.bfloat16()
    models.append(m)
px = [torch.randn(B, 3, 224, 224, device=dev).bfloat16() for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
N = int(os.environ.get("N", 40))


def seq():
    for i in range(N):
        models[i & 1].encode_image(px[i & 1], normalize=True)


def par():
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(cur)
    for i in range(N):
        with torch.cuda.stream(streams[i & 1]):
            models[i & 1].encode_image(px[i & 1], normalize=True)
    for s in streams:
        cur.wait_stream(s)


for name, fn in (("one stream", seq), ("two streams", par), ("one stream", seq), ("two streams", par)):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:12s}: {dt / N * 1e3:7.3f} ms per forward  {N * B / dt:9.0f} images/s", flush=True)
