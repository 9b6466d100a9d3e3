"""Per-kernel timeline of one factorisation from a rocprofv3 --kernel-trace csv: start offset, duration, gap to the
previous kernel's end (all streams merged).  usage: python scripts/chol_timeline.py <kernel_trace.csv> [fit index]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one fit starts with kmat_tile_kernel (K(X,X))
starts = [i for i, r in enumerate(rows) if "kmat_tile_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
seg = rows[starts[k]:(starts[k + 1] if k + 1 < len(starts) else len(rows))]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
for r in seg:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cbo::", "").replace("cbo::", "")[:34]
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:7.1f} us  gap {(a - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3}  {name}  grid {r.get('Grid_Size', '')}")
    prev_end = max(prev_end, b)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
