"""Latency of small-batch calls -- the reference's own call pattern is batch 1 (build_cache, reference
code/search_image.py:153-158) and batch ~10 (outlier_filter, :305-316): eager launches vs hipGraph replay, for
encode_image alone and for encode_image + top-10 search over a gallery, replay checked bit-identical to eager.
Prints one JSON object (profiles/r02_small_batch_latency.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmr_amd
from mmr_amd import search, synth

dev = torch.device("cuda:0")
name = os.environ.get("MODEL", "ViT-B/32")
model, _ = mmr_amd.load(name, device=dev, weights="synthetic")
S, E = model.input_resolution, model.cfg.embed_dim
index = search.GalleryIndex(synth.synth_unit_rows(100_000, E, seed=1).to(dev))
res = {"model": name, "gallery_rows": 100_000, "note": "ms per call, wall clock over 50 back-to-back calls; 1 x MI355X; seeded weights"}


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in [int(x) for x in os.environ.get("BS", "1,10,32").split(",")]:
    px = torch.randn(B, 3, S, S, device=dev)

    def enc():
        return model.encode_image(px, normalize=True)

    def enc_search():
        return index.search(model.encode_image(px, normalize=True), 10)

    row = {}
    for tag, fn in (("encode", enc), ("encode+top10", enc_search)):
        eager = timed(fn)
        ref = fn()
        ref = [t.clone() for t in (ref if isinstance(ref, tuple) else (ref,))]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                out = fn()
        torch.cuda.current_stream().wait_stream(side)
        out = out if isinstance(out, tuple) else (out,)
        replay = timed(g.replay)
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        row[tag] = {"eager_ms": round(eager, 4), "graph_replay_ms": round(replay, 4), "replay_equals_eager": bool(same),
                    "images_per_s_eager": round(B / eager * 1e3, 1), "images_per_s_graph": round(B / replay * 1e3, 1)}
    # independent requests in flight: one captured forward per LANE (own workspace, own stream), L of them replayed at once.
    # A small-batch forward is ~100 dependent launches that leave the chip nearly idle; eager launching is host-bound
    # (~3 us per launch), a graph replay is one host call, so the lanes overlap on the GPU.
    if B <= 10:
        fl = {}
        for lanes in (1, 2, 4, 8):
            streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
            graphs, outs = [], []
            for l in range(lanes):
                with torch.cuda.stream(streams[l]):
                    model.encode_image(px, normalize=True, lane=l)
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg, stream=streams[l]):
                        outs.append(model.encode_image(px, normalize=True, lane=l))
                graphs.append(gg)
            torch.cuda.synchronize()

            def burst(n=400):
                for i in range(n):
                    with torch.cuda.stream(streams[i % lanes]):
                        graphs[i % lanes].replay()
            burst(50)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            burst()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 400 * 1e3
            ref1 = model.encode_image(px, normalize=True)
            fl[str(lanes)] = {"ms_per_forward": round(ms, 4), "images_per_s": round(B / ms * 1e3, 1),
                              "equals_eager": bool(all(torch.equal(o, ref1) for o in outs))}
        row["encode_graphs_in_flight"] = fl
    res[f"batch_{B}"] = row
print(json.dumps(res, indent=1))
