import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = _lib.stream_ptr(dev)
M, N, K = 12800, 3072, 768
A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
for epi, name in [(0, "bias->bf16"), (1, "quickgelu->bf16"), (5, "gelu-erf->bf16"), (4, "bias->f32"), (3, "plain f32")]:
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
    for _ in range(5): L.mmr_debug_gemm(epi, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): L.mmr_debug_gemm(epi, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), st)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    print(f"fc1-shape epi {name:16s}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
