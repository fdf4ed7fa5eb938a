"""A/B aid: forward time of the towers with the library named by MMR_LIB (default: the in-tree build).
    MMR_LIB=tools/_ab/lib_x.so python tools/ab_forward.py [cfg2,cfg5,text]
Prints one JSON line: wall ms per forward (HIP events around `reps` back-to-back forwards) per workload and a checksum of
the features (two builds that claim bit-identity must print the same one)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import synth

dev = torch.device("cuda:0")
which = (sys.argv[1] if len(sys.argv) > 1 else "cfg2,cfg5,text").split(",")
reps = int(os.environ.get("REPS", "20"))
out = {"lib": os.environ.get("MMR_LIB", "in-tree")}


def time_fn(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, r


for w in which:
    if w in ("cfg2", "cfg5", "b512"):
        name, B = {"cfg2": ("ViT-B/32", 256), "cfg5": ("ViT-L/14@336px", 128), "b512": ("ViT-B/32", 512)}[w]
        model, _ = mmr_amd.load(name, device=dev, weights="synthetic")
        model.bfloat16()
        S = model.input_resolution
        px = torch.randn(B, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).bfloat16()
        ms, f = time_fn(lambda: model.encode_image(px, normalize=True))
    else:
        model, _ = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
        model.bfloat16()
        ids = synth.synth_token_ids(256, 77, model.cfg.text.vocab, seed=5).to(dev)
        ms, f = time_fn(lambda: model.encode_text(ids, normalize=True))
    out[w] = {"ms": round(ms, 4), "checksum": float(f.double().abs().sum()), "finite": bool(torch.isfinite(f.float()).all())}
    del model
    torch.cuda.empty_cache()
print(json.dumps(out))
