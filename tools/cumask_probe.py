"""Probe: CU-masked streams (hipExtStreamCreateWithCUMask) -- does a mask restrict a kernel to a CU subset, and
do two half-chip streams overlap a memory-bound and a compute-bound kernel?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import _lib

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init()
L = _lib.lib()

def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return st

M, N, K = 12800, 2304, 768
A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
torch.cuda.synchronize()

def time_on(stream_ptr, iters=20):
    ext = torch.cuda.ExternalStream(stream_ptr.value if isinstance(stream_ptr, ctypes.c_void_p) else stream_ptr)
    with torch.cuda.stream(ext):
        for _ in range(3):
            L.mmr_debug_gemm(0, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), ext.cuda_stream)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(ext)
        for _ in range(iters):
            L.mmr_debug_gemm(0, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), ext.cuda_stream)
        e.record(ext)
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3

full = masked_stream(range(256))
print("all 256 bits   :", round(time_on(full), 1), "us")
for name, bits in [("bits 0-127", range(0, 128)), ("bits 128-255", range(128, 256)), ("even bits", range(0, 256, 2)),
                   ("bits 0-63", range(0, 64)), ("bits 0-31", range(0, 32)), ("every 8th..+3", [b for b in range(256) if b % 8 < 4])]:
    st = masked_stream(bits)
    print(f"{name:15s}:", round(time_on(st), 1), "us")
