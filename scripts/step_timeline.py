"""One overlapped step (cbo_gp_fit_sweep) from a rocprofv3 --kernel-trace csv: per-stream timeline and a summary of what
the chain stream spent where (kernel time by kind, waits between its launches), when each stream finished, and the
closing launch.  usage: step_timeline.py <dir or csv> [--step K] [-q]   (K-th step from the end, default 2)"""
import csv, sys, glob, os

arg = sys.argv[1]
path = arg if arg.endswith(".csv") else sorted(glob.glob(os.path.join(arg, "**", "*_kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "rhs_kernel" in r["Kernel_Name"]]
back = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 2
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = rows[i0]["s"]
quiet = "-q" in sys.argv


def short(n):
    for k in ("potrf_panel_fused", "potrf_diag128", "syrk_rows", "syrk_kernel", "trsm_update", "trsm_strip8", "trsm_strip_kernel<true",
              "trsm_strip_kernel<false", "panel_trsm", "kmat", "acq_kernel", "argmax", "rhs", "zero_ints"):
        if k in n:
            return k
    return n.split("(")[0][-30:]


step = [r for r in rows[i0:i1]]
chain_q = rows[i0]["Queue_Id"]
by_kind, waits, last_end = {}, 0.0, None
chain_end = None
q_end = {}
for r in step:
    k = short(r["Kernel_Name"])
    q = r["Queue_Id"]
    if not quiet:
        print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:8.1f} q{q} {k} grid={r['Grid_Size_X']}x{r.get('Grid_Size_Y','')}")
    if k == "trsm_strip8":
        closing = r
        continue
    if k in ("acq_kernel", "argmax", "kmat") and r["s"] > rows[i0]["s"] + 1000000:
        continue
    q_end[q] = max(q_end.get(q, 0), (r["e"] - t0) / 1e3)
    if q == chain_q and k in ("potrf_panel_fused", "potrf_diag128", "syrk_rows", "panel_trsm"):
        by_kind[k] = by_kind.get(k, 0.0) + (r["e"] - r["s"]) / 1e3
        if last_end is not None:
            waits += max(0.0, (r["s"] - last_end) / 1e3)
        last_end = r["e"]
        chain_end = (r["e"] - t0) / 1e3
print("chain kernels (us):", {k: round(v, 1) for k, v in by_kind.items()}, " waits between them:", round(waits, 1), " chain ends at", round(chain_end, 1))
print("stream ends (us):", {f"q{q}": round(v, 1) for q, v in sorted(q_end.items())})
try:
    print(f"closing launch: {(closing['s']-t0)/1e3:.1f} -> {(closing['e']-t0)/1e3:.1f}  ({(closing['e']-closing['s'])/1e3:.1f} us)")
except NameError:
    pass
