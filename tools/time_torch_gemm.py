"""Calibration only: what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the encoder's shapes."""
import torch
dev = torch.device("cuda:0")
for name, M, N, K in [("qkv", 12800, 2304, 768), ("out", 12800, 768, 768), ("fc1", 12800, 3072, 768), ("fc2", 12800, 768, 3072)]:
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    for _ in range(5):
        torch.nn.functional.linear(A, W, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        torch.nn.functional.linear(A, W, b)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    print(f"vendor {name:4s} {M}x{N}x{K}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
