"""Throughput of the on-device preprocess (development aid): 256 uint8 HWC images -> [256,3,224,224] bf16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmr_amd import preprocess

dev = torch.device("cuda:0")
H, W, B = int(os.environ.get("H", 480)), int(os.environ.get("W", 640)), int(os.environ.get("B", 256))
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev) for _ in range(B)]
for fn, name in ((lambda: preprocess.preprocess_batch(imgs, 224, torch.bfloat16), "preprocess_batch (one launch pair)"),
                 (lambda: [preprocess.preprocess_image(im, 224, out_dtype=torch.bfloat16) for im in imgs], "preprocess_image x B")):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        keep = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:36s} {H}x{W} x{B}: {ms:8.3f} ms  {B / ms * 1e3:9.0f} images/s  ({B * H * W * 3 / ms / 1e6:.1f} GB/s of uint8 in)", flush=True)
