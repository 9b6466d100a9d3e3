"""Summarise rocprofv3 --pmc csv output per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections, re
root = sys.argv[1]
for f in sorted(glob.glob(root + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"])[:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f)
    for k, cs in acc.items():
        n = max(len(v) for v in cs.values())
        print(f"  {k}  dispatches={n}")
        for c, v in sorted(cs.items()):
            print(f"      {c:34s} mean={sum(v)/len(v):.6g}  max={max(v):.6g}")
