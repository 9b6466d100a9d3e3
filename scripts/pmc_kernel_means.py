"""Per-launch means of the counters of one rocprofv3 --pmc run for the kernels whose name contains a pattern.
usage: python scripts/pmc_kernel_means.py <output dir of rocprofv3 -d> <pattern>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    if sys.argv[2] in k:
        print(k, max(n[k].values()), "launches:", {c: round(x / n[k][c]) for c, x in v.items()})
