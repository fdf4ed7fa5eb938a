"""Quick timing of the encode path on the GPU box (development aid; bench.py is the contract)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd

dev = torch.device("cuda:0")
name = os.environ.get("MODEL", "ViT-B/32")
B = int(os.environ.get("B", 256))
model, _ = mmr_amd.load(name, device=dev, weights="synthetic")
model.bfloat16()
S = model.input_resolution
px = torch.randn(B, 3, S, S, device=dev).bfloat16()
for _ in range(3):
    model.encode_image(px, normalize=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = int(os.environ.get("ITERS", 10))
s.record()
for _ in range(iters):
    model.encode_image(px, normalize=True)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
gflop = {"ViT-B/32": 8.8176, "ViT-L/14": 162.03, "ViT-L/14@336px": 381.92}[name]
print(f"{name} B={B}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} img/s  {B*gflop/ms:.1f} TFLOP/s", flush=True)
if os.environ.get("TEXT"):
    ids = mmr_amd.synth.synth_token_ids(B, 77, model.cfg.text.vocab).to(dev)
    for _ in range(3):
        model.encode_text(ids, normalize=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        model.encode_text(ids, normalize=True)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"text tower B={B}: {ms:.3f} ms/batch  {B/ms*1e3:.0f} texts/s", flush=True)
