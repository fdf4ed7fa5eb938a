"""Times the attention core alone through mmr_debug_attention (development aid).  B, T, HEADS via env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import _lib

dev = torch.device("cuda:0")
B, T, H = int(os.environ.get("B", 128)), int(os.environ.get("T", 577)), int(os.environ.get("HEADS", 16))
causal = int(os.environ.get("CAUSAL", 0))
d = H * 64
qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).bfloat16()
o = torch.empty(B * T, d, dtype=torch.bfloat16, device=dev)
L, st = _lib.lib(), _lib.stream_ptr(dev)
it = int(os.environ.get("ITERS", 20))
for _ in range(3):
    _lib.check(L.mmr_debug_attention(qkv.data_ptr(), o.data_ptr(), B, T, H, causal, st))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(it):
    L.mmr_debug_attention(qkv.data_ptr(), o.data_ptr(), B, T, H, causal, st)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / it * 1e3
flop = 4.0 * B * H * T * T * 64
print(f"attention B={B} T={T} heads={H}: {us:.1f} us  {flop / us / 1e6:.0f} TFLOP/s", flush=True)
