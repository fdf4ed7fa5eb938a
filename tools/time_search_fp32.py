"""Timing of the search over fp32 galleries (development aid).  NS / QS: comma-separated gallery sizes / query counts;
MMR_SCAN_F32=exact selects the fp32-MFMA scan instead of the split-bf16 one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import search
dev = torch.device("cuda:0")
for N in [int(x) for x in os.environ.get("NS", "10000,100000,1000000").split(",")]:
    g = torch.randn(N, 512, device=dev); g = g / g.norm(dim=-1, keepdim=True)
    idx = search.GalleryIndex(g)
    for Q in [int(x) for x in os.environ.get("QS", "1,16,128").split(",")]:
        q = torch.randn(Q, 512, device=dev); q = q / q.norm(dim=-1, keepdim=True)
        for _ in range(2): idx.search(q, 10)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): idx.search(q, 10)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print(f"fp32 N={N:8d} Q={Q:4d}: {ms:9.3f} ms  {Q*N/ms/1e6:8.2f} Gpairs/s", flush=True)
