"""Micro-benchmark of gemm_bf16_kernel on the ViT-B/32 B=256 shapes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_ptr(dev)
shapes = [("qkv", 0, 12800, 2304, 768), ("out", 2, 12800, 768, 768), ("fc1", 1, 12800, 3072, 768),
          ("fc2", 2, 12800, 768, 3072), ("patch", 3, 12544, 768, 3072)]
if os.environ.get("SHAPES"):      # "name:epi:M:N:K,..."
    shapes = [(a, int(b), int(c), int(d), int(e)) for a, b, c, d, e in (x.split(":") for x in os.environ["SHAPES"].split(","))]
tot = 0.0
for name, epi, M, N, K in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    if os.environ.get("ZEROS") == "1":       # DVFS check: same instructions on all-zero operands (MI355X_MICROARCH 'DVFS give-back' item 1)
        A.zero_(); W.zero_()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi >= 2 else torch.bfloat16)
    for _ in range(5):
        _lib.check(L.mmr_debug_gemm(epi, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), st))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 50
    s.record()
    for _ in range(iters):
        L.mmr_debug_gemm(epi, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), st)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    if name != "patch":
        tot += us
    print(f"{name:6s} {M}x{N}x{K}: {us:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
print(f"per-layer GEMM sum: {tot:.1f} us")
