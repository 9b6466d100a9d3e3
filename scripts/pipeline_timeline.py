"""Timeline of one bench step from a rocprofv3 --kernel-trace csv: start/end/duration per kernel and queue.
usage: pipeline_timeline.py trace.csv [--step K] [-q]   (K-th step from the end; bench.py appends 4 two-call steps)"""
import csv, sys, glob
path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/ovtrace*/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "rhs_kernel" in r["Kernel_Name"]]
back = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 1     # 1 = last step of the run
i0 = starts[-back]
i1 = starts[-back + 1] - 3 if back > 1 else len(rows)
t0 = rows[i0]["s"]
def short(n):
    for k in ("potrf_diag128", "syrk", "trsm_update", "trsm_strip_kernel<true", "trsm_strip_kernel<false", "kmat", "acq_kernel", "argmax", "rhs", "prep"):
        if k in n:
            return k
    return n[:30]
busy = {}
for r in rows[i0 - 3:i1]:
    k = short(r["Kernel_Name"])
    busy[k] = busy.get(k, 0) + (r["e"] - r["s"]) / 1e3
    if "-q" not in sys.argv:
        print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:8.1f} q{r['Queue_Id']} {k} grid={r['Grid_Size_X']}x{r.get('Grid_Size_Y','')}")
print({k: round(v, 1) for k, v in busy.items()})
