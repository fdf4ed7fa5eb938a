"""Experiment: split the batch over S streams so memory-bound phases of one slice overlap MFMA phases of another."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd.clip import _Tower
from mmr_amd import weights

dev = torch.device("cuda:0")
cfg = mmr_amd.get_config("ViT-B/32")
w = weights.make_vision_weights(cfg.vision)
B = 256
px = torch.randn(B, 3, 224, 224, device=dev).bfloat16()
for S in (1, 2, 3, 4):
    towers = [_Tower(cfg.vision, w, dev) for _ in range(S)]   # separate workspaces (weights duplicated: experiment only)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    bounds = [B * i // S for i in range(S + 1)]
    def run():
        outs = []
        cur = torch.cuda.current_stream(dev)
        for i in range(S):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(towers[i].forward(px[bounds[i]:bounds[i + 1]], torch.bfloat16, True))
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
    print(f"streams={S}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} img/s", flush=True)
    del towers
