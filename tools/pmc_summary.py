"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM bytes per kernel.

    python tools/pmc_summary.py <fetch_dir> <write_dir> [out.json]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes:
  * FETCH_SIZE / WRITE_SIZE are in KiB (bytes = value * 1024);
  * on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
    (16 B/lane global_load and buffer_load ... lds alike) -> the read side is doubled;
  * WRITE_SIZE reads exactly for 16-B-per-lane stores; other widths are uncalibrated.
"""
import csv, glob, json, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("mmr::", "").split("(")[0]
            a = acc[name]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
rows = {}
for name in sorted(set(fetch) | set(write)):
    if not any(k in name for k in ("scan_kernel", "gemm", "select_kernel",
                                   "rescore_kernel", "rank_kernel", "attention_kernel", "layernorm_kernel",
                                   "im2col", "embed_")):
        continue
    f, fn = fetch.get(name, (0.0, 0))
    w, wn = write.get(name, (0.0, 0))
    rd = 2.0 * f * 1024 / max(fn, 1)          # gfx950: FETCH_SIZE counts 64 B per 128-B request
    wr = w * 1024 / max(wn, 1)
    rows[name] = {"launches": fn, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr),
                  "raw_FETCH_SIZE_KiB_per_launch": round(f / max(fn, 1), 1),
                  "raw_WRITE_SIZE_KiB_per_launch": round(w / max(wn, 1), 1)}
    print(f"{name[:48]:48s} n={fn:5d} read {rd/1e6:10.2f} MB  write {wr/1e6:9.2f} MB per launch")
out = {"note": "per-launch HBM bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); "
               "read side doubled per the gfx950 correction in MI355X_MICROARCH.md, which is calibrated for "
               "wide (16 B/lane) coalesced streams -- scan_kernel and the GEMM global_load_lds staging; for the "
               "gather-style kernels (im2col, attention, select) it over-states reads by up to 2x",
       "kernels": rows}
gemm = [v for k, v in rows.items() if k.startswith("gemm")]
if gemm:
    tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in gemm)
    n = sum(v["launches"] for v in gemm)
    out["gemm_bytes_per_launch"] = round(tot / max(n, 1))
scan = [v for k, v in rows.items() if k.startswith("scan_kernel")]
if scan:
    out["scan_bytes_per_launch"] = scan[0]["hbm_bytes_per_launch"]
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
