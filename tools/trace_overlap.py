"""How much of a two-lane bench run actually overlaps: reads a rocprofv3 --kernel-trace CSV of `python bench.py` (two steps in
flight during the timed region), takes the kernels of the two lane queues in the last 60 % of their window and prints wall time,
summed kernel time, time with >= 1 and >= 2 library kernels running, and the summed time per kernel class.
    python tools/trace_overlap.py gpurun_out/prof_bench2"""
import collections, csv, glob, sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_bench2"
files = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=lambda p: -len(open(p).read()))
rows = [r for r in csv.DictReader(open(files[0])) if "mmr::" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
byq = collections.Counter(r["Queue_Id"] for r in rows)
lanes = [q for q, _ in byq.most_common()][1:]            # the busiest queue is the caller's stream (one-step-in-flight passes)
lr = [r for r in rows if r["Queue_Id"] in lanes]
t0, t1 = min(r["s"] for r in lr), max(r["e"] for r in lr)
sel = [r for r in lr if r["s"] >= t0 + (t1 - t0) * 0.4]
wall = max(r["e"] for r in sel) - min(r["s"] for r in sel)
ev = sorted([(r["s"], 1) for r in sel] + [(r["e"], -1) for r in sel])
cur, last, one, two = 0, None, 0, 0
for t, dlt in ev:
    if cur > 0:
        one += t - last
    if cur > 1:
        two += t - last
    cur += dlt
    last = t
cls = collections.defaultdict(float)
for r in sel:
    n = r["Kernel_Name"]
    k = ("gemm" if "gemm" in n else "layernorm" if "layernorm" in n else "attention" if "attention" in n
         else "scan" if "scan_kernel" in n else "other")
    cls[k] += (r["e"] - r["s"]) / 1e6
print(f"kernels of the lane queues {byq.most_common()[1:]} in the last 60 % of their window: {len(sel)}")
print(f"wall {wall / 1e6:.2f} ms | sum of kernel durations {sum(r['e'] - r['s'] for r in sel) / 1e6:.2f} ms | "
      f">= 1 kernel running {one / 1e6:.2f} ms ({100 * one / wall:.1f} %) | >= 2 running {two / 1e6:.2f} ms ({100 * two / wall:.1f} %)")
print("summed durations by class (ms):", {k: round(v, 2) for k, v in sorted(cls.items())})
