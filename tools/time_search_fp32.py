"""Timing of the exact (exhaustive fp64) path that fp32 galleries take (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import search
dev = torch.device("cuda:0")
for N in (10_000, 100_000, 1_000_000):
    g = torch.randn(N, 512, device=dev); g = g / g.norm(dim=-1, keepdim=True)
    idx = search.GalleryIndex(g)
    for Q in (1, 16, 128):
        q = torch.randn(Q, 512, device=dev); q = q / q.norm(dim=-1, keepdim=True)
        for _ in range(2): idx.search(q, 10)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): idx.search(q, 10)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print(f"fp32 N={N:8d} Q={Q:4d}: {ms:9.3f} ms  {Q*N/ms/1e6:8.2f} Gpairs/s", flush=True)
