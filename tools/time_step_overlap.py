"""Does running the step's two independent legs (encode 256 images; top-10 of 256 queries over 1M rows) on two HIP
streams beat running them back to back?  (development aid; bench.py is the contract)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import search, synth

dev = torch.device("cuda:0")
model, _ = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
model.bfloat16()
px = torch.randn(256, 3, 224, 224, device=dev).bfloat16()
gal = torch.randn(1_000_000, 512, device=dev)
gal = (gal / gal.norm(dim=-1, keepdim=True)).bfloat16()
q = synth.synth_unit_rows(256, 512, seed=4).bfloat16().to(dev)
index = search.GalleryIndex(gal)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def seq():
    f = model.encode_image(px, normalize=True)
    o = index.search(q, 10, 1.0)
    return f, o


def par():
    cur = torch.cuda.current_stream(dev)
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        o = index.search(q, 10, 1.0)
    with torch.cuda.stream(s1):
        f = model.encode_image(px, normalize=True)
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return f, o


for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        keep = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:12s}: {ms:.3f} ms/step  {256 / ms * 1e3:.0f} images/s", flush=True)
fa, oa = seq()
fb, ob = par()
torch.cuda.synchronize()
print("same results:", torch.equal(fa, fb), torch.equal(oa[1], ob[1]), torch.equal(oa[0], ob[0]))
