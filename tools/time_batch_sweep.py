"""Encode latency / throughput across batch sizes (development aid): ViT-B/32 bf16 pixels, eager calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd

dev = torch.device("cuda:0")
model, _ = mmr_amd.load(os.environ.get("MODEL", "ViT-B/32"), device=dev, weights="synthetic")
model.bfloat16()
S = model.input_resolution
for B in [int(x) for x in os.environ.get("BS", "1,8,16,32,48,64,96,128,192,256").split(",")]:
    px = torch.randn(B, 3, S, S, device=dev).bfloat16()
    for _ in range(3):
        model.encode_image(px)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        model.encode_image(px)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:4d}  {ms:7.3f} ms  {B / ms * 1e3:9.0f} img/s  {B * 8.8176 / ms:7.1f} TFLOP/s", flush=True)
