"""Stress aid: the encoder and the search must return bit-identical results run after run, also while another stream
keeps the GPU busy (timing-dependent bugs in hand-counted waits would show up as run-to-run differences)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import search, synth

dev = torch.device("cuda:0")
model, _ = mmr_amd.load(os.environ.get("MODEL", "ViT-B/32"), device=dev, weights="synthetic")
model.bfloat16()
B = int(os.environ.get("B", 256))
S = model.input_resolution
px = torch.randn(B, 3, S, S, device=dev).bfloat16()
ids = synth.synth_token_ids(B, 77, model.vocab_size, seed=5).to(dev)
gal = torch.randn(300_000, model.cfg.embed_dim, device=dev)
gal = (gal / gal.norm(dim=-1, keepdim=True)).bfloat16()
q = gal[:200].clone()
index = search.GalleryIndex(gal)
ref_img = model.encode_image(px, normalize=True).clone()
ref_txt = model.encode_text(ids, normalize=True).clone()
ref_s = [t.clone() for t in index.search(q, 10, 1.0)]
# the other scan kernels: E = 768 bf16 (scan16_kernel) and fp32 galleries (scan_f32s_kernel: LDS-DMA ring + split images)
extra = []
for E, dt, nq in ((768, torch.bfloat16, 128), (512, torch.float32, 128), (768, torch.float32, 64), (128, torch.float32, 100),
                  (256, torch.float32, 33)):
    g2 = torch.randn(200_003, E, device=dev)
    g2 = (g2 / g2.norm(dim=-1, keepdim=True)).to(dt)
    q2 = g2[5:5 + nq].clone()
    ix = search.GalleryIndex(g2)
    extra.append((ix, q2, [t.clone() for t in ix.search(q2, 10, 1.0, return_dot64=True, return_status=True)]))
side = torch.cuda.Stream(dev)
big = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
bad = 0
iters = int(os.environ.get("ITERS", 40))
for it in range(iters):
    if it % 2:                      # every other iteration: contend with a torch GEMM + copy on a side stream
        with torch.cuda.stream(side):
            for _ in range(3):
                big @ big
                big.clone()
    a = model.encode_image(px, normalize=True)
    t = model.encode_text(ids, normalize=True)
    s = index.search(q, 10, 1.0)
    bad += int(not torch.equal(a, ref_img)) + int(not torch.equal(t, ref_txt))
    bad += int(not (torch.equal(s[0], ref_s[0]) and torch.equal(s[1], ref_s[1])))
    for ix, q2, ref in extra:
        got = ix.search(q2, 10, 1.0, return_dot64=True, return_status=True)
        bad += int(not all(torch.equal(a_, b_) for a_, b_ in zip(got, ref)))     # ids, scores, fp64 dots AND the path status
torch.cuda.synchronize()
print(f"{iters} iterations, mismatches: {bad}")
sys.exit(1 if bad else 0)
