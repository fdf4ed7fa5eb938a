"""Experiment: split the batch over S CU-masked streams (disjoint CU ranges), so the memory-bound phases of one
slice overlap the MFMA-bound phases of another."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd.clip import _Tower
from mmr_amd import weights

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init()

def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)

cfg = mmr_amd.get_config("ViT-B/32")
w = weights.make_vision_weights(cfg.vision)
B = 256
px = torch.randn(B, 3, 224, 224, device=dev).bfloat16()
base = _Tower(cfg.vision, w, dev)
for S in (1, 2, 4):
    towers = [_Tower(cfg.vision, w, dev) for _ in range(S)]
    streams = [masked_stream(256 * i // S, 256 * (i + 1) // S) for i in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    bounds = [B * i // S for i in range(S + 1)]
    def run():
        outs = []
        cur = torch.cuda.current_stream(dev)
        for i in range(S):
            if S > 1:
                streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(towers[i].forward(px[bounds[i]:bounds[i + 1]], torch.bfloat16, True))
        if S > 1:
            for i in range(S):
                cur.wait_stream(streams[i])
        return outs
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"cu-masked streams={S}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} img/s", flush=True)
    del towers
