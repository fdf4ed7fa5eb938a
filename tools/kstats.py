"""Print a compact per-kernel summary from a rocprofv3 *_kernel_stats.csv (names truncated)."""
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        n = r["Name"].replace("void ", "").replace("mmr::", "")
        n = n[:60]
        print(f"{n:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} "
              f"{float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
