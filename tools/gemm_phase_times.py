"""Phase breakdown of the 256-row GEMM kernels from in-kernel s_memrealtime stamps (development aid).

Needs a DIAGNOSTIC build of the library with -DMMR_GEMM_STAMPS (never the shipped one):
    MMR_EXTRA_HIPCC_FLAGS=-DMMR_GEMM_STAMPS python -m mmr_amd.csrc.build --force   (into a copy: MMR_LIB=...)
    python tools/gemm_phase_times.py
Per workgroup (wave 0) and tile: t0 tile start, t1 first K-tile landed, t2 main loop done, t3 epilogue issued.
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mmr_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_ptr(dev)
L.mmr_debug_gemm_stamps.restype = ctypes.c_int
L.mmr_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
shapes = [("qkv", 0, 12800, 2304, 768), ("fc1", 1, 12800, 3072, 768), ("out", 2, 12800, 768, 768), ("fc2", 2, 12800, 768, 3072)]
if os.environ.get("SHAPES") == "resid":       # the fp32-residual launches of ViT-B/32 batch 256 / 512 and ViT-L/14@336 batch 128
    shapes = [("out", 2, 12800, 768, 768), ("fc2", 2, 12800, 768, 3072), ("out-b512", 2, 25600, 768, 768),
              ("L14-out", 2, 73984, 1024, 1024), ("L14-fc2", 2, 73984, 1024, 4096)]
for name, epi, M, N, K in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi >= 2 else torch.bfloat16)
    for _ in range(max(50, int(os.environ.get("WARM", 2000)) * 12800 // M)):       # long enough for the clock to settle under this load
        _lib.check(L.mmr_debug_gemm(epi, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), st))
    torch.cuda.synchronize()
    L.mmr_debug_gemm_stamps(None, 1)
    _lib.check(L.mmr_debug_gemm(epi, A.data_ptr(), W.data_ptr(), M, N, K, bias.data_ptr(), out.data_ptr(), st))
    torch.cuda.synchronize()
    buf = np.zeros(2 * 8192 * 4, dtype=np.uint64)
    L.mmr_debug_gemm_stamps(buf.ctypes.data, 0)
    s = buf[:8192 * 4].reshape(-1, 4).astype(np.int64)
    cyc = buf[8192 * 4:].reshape(-1, 4).astype(np.int64)
    live = s[:, 0] > 0
    s, cyc = s[live], cyc[live]
    ghz = (cyc[:, 2] - cyc[:, 1]) / ((s[:, 2] - s[:, 1]) * 10.0)     # cycles per ns over the main loop
    print(f"   in-kernel clock over the main loop: median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz,10):.3f}, p90 {np.percentile(ghz,90):.3f})")
    t00 = s[:, 0].min()
    rel = (s - t00) / 100.0                      # us since the first workgroup started
    d = np.diff(rel, axis=1)
    print(f"{name} {M}x{N}x{K}: {len(s)} tiles, kernel span {rel.max():.1f} us")
    print(f"   prologue wait  mean {d[:,0].mean():6.2f}  p10 {np.percentile(d[:,0],10):6.2f}  p90 {np.percentile(d[:,0],90):6.2f} us")
    print(f"   main loop      mean {d[:,1].mean():6.2f}  p10 {np.percentile(d[:,1],10):6.2f}  p90 {np.percentile(d[:,1],90):6.2f} us")
    print(f"   epilogue issue mean {d[:,2].mean():6.2f}  p10 {np.percentile(d[:,2],10):6.2f}  p90 {np.percentile(d[:,2],90):6.2f} us")
    order = np.argsort(rel[:, 0])
    starts = rel[order, 0]
    print("   tile start times (us), every 32nd in start order:", np.round(starts[::32], 1).tolist())
    ends = np.sort(rel[:, 3])
    print("   tile end times   (us), every 32nd in end order:  ", np.round(ends[::32], 1).tolist())
