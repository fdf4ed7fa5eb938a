"""One-GPU timings of the BASELINE.json configs that bench.py's headline line does not cover (bench.py measures
configs[1]; this writes one JSON object for configs[2], the per-GPU share of configs[3], and configs[4]).
Inputs are synthetic and device-resident, weights seeded, like bench.py.

    python tools/bench_other_configs.py > profiles/r01_other_configs.json
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import search, synth

dev = torch.device("cuda:0")


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        keep = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def unit_rows(n, e, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty(n, e, dtype=torch.bfloat16, device=dev)
    for s in range(0, n, 1 << 18):
        x = torch.randn(min(n, s + (1 << 18)) - s, e, generator=g, device=dev)
        out[s:s + x.shape[0]] = (x / x.norm(dim=-1, keepdim=True)).bfloat16()
    return out


res = {"note": "1 x MI355X, synthetic device-resident inputs, seeded random-init weights; ms are wall-clock means"}

# configs[2]: text-to-image -- CLIP text tower bf16 (ViT-B/32's) on 256 prompts, their features as the queries of a
# top-10 search over a 1M x 512 image gallery
model, _ = mmr_amd.load("ViT-B/32", device=dev, weights="synthetic")
model.bfloat16()
ids = synth.synth_token_ids(256, 77, model.vocab_size, seed=5).to(dev)
gal = unit_rows(1_000_000, 512, 3)
index = search.GalleryIndex(gal)


def text_to_image():
    f = model.encode_text(ids, normalize=True)
    return index.search(f, 10, 1.0)


t_enc = timed(lambda: model.encode_text(ids, normalize=True), 20)
t_all = timed(text_to_image, 20)
res["configs[2] text-to-image"] = {"texts": 256, "gallery": "1000000x512 bf16", "k": 10, "encode_text_ms": round(t_enc, 3),
                                   "encode_plus_search_ms": round(t_all, 3), "texts_per_s": round(256 / t_enc * 1e3),
                                   "end_to_end_queries_per_s": round(256 / t_all * 1e3)}

# configs[3]: 10M-image gallery over 8 GPUs -> 1.25M rows per GPU; the local top-10 each rank computes before the
# all-gather (the collective itself is a 40 KB/rank exchange, covered by tests and bench.py --gpus N)
gal = None
index = None
torch.cuda.empty_cache()
shard = unit_rows(1_250_000, 512, 7)
q = unit_rows(256, 512, 8)
index = search.GalleryIndex(shard)
t = timed(lambda: index.search(q, 10, 1.0), 20)
res["configs[3] per-GPU shard of a 10M gallery"] = {"rows": 1_250_000, "queries": 256, "k": 10, "search_ms": round(t, 3),
                                                      "gpairs_per_s": round(256 * 1.25e6 / t * 1e3 / 1e9, 1)}
shard = index = None
model = None
torch.cuda.empty_cache()

# configs[4]: ViT-L/14@336 bf16, batch 128 per GPU, + search over a 768-d gallery
model, _ = mmr_amd.load("ViT-L/14@336px", device=dev, weights="synthetic")
model.bfloat16()
px = torch.randn(128, 3, 336, 336, device=dev).bfloat16()
gal = unit_rows(1_000_000, 768, 9)
index = search.GalleryIndex(gal)
q = unit_rows(128, 768, 10)
t_enc = timed(lambda: model.encode_image(px, normalize=True), 5, warm=2)
t_s = timed(lambda: index.search(q, 10, 1.0), 20)
res["configs[4] ViT-L/14@336"] = {"batch": 128, "encode_ms": round(t_enc, 2), "images_per_s": round(128 / t_enc * 1e3),
                                  "encode_tflops": round(128 * 381.92 / t_enc, 1),
                                  "search_1Mx768_q128_ms": round(t_s, 3),
                                  "search_gpairs_per_s": round(128 * 1e6 / t_s * 1e3 / 1e9, 1)}
print(json.dumps(res, indent=1))
