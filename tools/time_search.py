"""Quick timing of the search path on the GPU box (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import search

dev = torch.device("cuda:0")
N, E = int(os.environ.get("N", 1_000_000)), int(os.environ.get("E", 512))
g = torch.randn(N, E, device=dev)
g = (g / g.norm(dim=-1, keepdim=True)).bfloat16()
idx = search.GalleryIndex(g)
for Q in [int(x) for x in os.environ.get("QS", "1,32,64,128,256,1024").split(",")]:
    q = torch.randn(Q, E, device=dev)
    q = (q / q.norm(dim=-1, keepdim=True)).bfloat16()
    for _ in range(3):
        idx.search(q, 10)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    s.record()
    for _ in range(iters):
        idx.search(q, 10)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    gb = N * E * 2 / 1e9
    print(f"Q={Q:5d}  {ms:8.3f} ms/batch  {Q/ms*1e3:12.0f} queries/s  {Q*N/ms*1e3/1e9:9.1f} Gpairs/s  "
          f"gallery-stream {gb*((Q+255)//256)/ms*1e3:7.1f} GB/s", flush=True)
