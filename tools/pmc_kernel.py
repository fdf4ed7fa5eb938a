"""Average the counters of a rocprofv3 --pmc run for kernels whose name contains a pattern (development aid).
    python tools/pmc_kernel.py <dir> <pattern>"""
import collections, csv, glob, sys
acc, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k] / n[k]:.4g}   (n={n[k]})")
