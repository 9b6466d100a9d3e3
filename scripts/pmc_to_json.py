"""HBM traffic from the rocprofv3 --pmc passes (scripts/pmc_passes.sh output) into profiles/pmc_traffic.json,
which bench.py reads for roofline.traffic:
  "step"             : FETCH_SIZE / WRITE_SIZE (KB) summed over the kernel dispatches of the run's last three steps
                       (bench.py --steps 2 --warmup 1 --post-steps 0; a step ends with argmax_final_kernel; the calls on
                       which cbo_gp_fit_sweep settles its schedule come before and are left out), divided by three
  "step_sequential"  : the same for --sequential
  "strip_kernel"     : mean per dispatch of trsm_pair_kernel<true> (rounds 3-4: trsm_strip8_kernel<true>) from the
                       --sequential passes
  "f32_strip_kernel" : mean per dispatch of trsm_strip_f32_kernel from the --dtype f32 passes
The file carries the hash of the kernel sources the passes ran on (bench.kernel_sources_sha): bench.py reports the
traffic only while that hash matches the tree.  Counter collection serialises the kernels; bytes per kernel do not
depend on that.
usage: python scripts/pmc_to_json.py <passes dir> <out json> [steps]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
def values(run, counter, pred, last_steps=None):
    """counter values of the run's dispatches; last_steps: only the dispatches of the run's last so many steps (a step
    ends with argmax_final_kernel) -- cbo_gp_fit_sweep settles its schedule on calls that come BEFORE bench.py's warm-up
    and run other schedules, and those are not the step's traffic"""
    rows = []
    for f in glob.glob(f"{root}/{run}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"].startswith(counter):
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    if last_steps is not None:
        ends = [i for i, r in enumerate(rows) if "argmax_final_kernel" in r[1]]
        if len(ends) > last_steps:
            rows = rows[ends[-last_steps - 1] + 1:]
    return [v for _, name, v in rows if pred(name)]
res = {"step": {}, "step_sequential": {}, "strip_kernel": {}, "f32_strip_kernel": {}}
for run, counter, key in (("fetch", "FETCH_SIZE", "fetch_size_kb"), ("write", "WRITE_SIZE", "write_size_kb")):
    v = values(run, counter, lambda n: True, steps)
    res["step"][key] = sum(v) / steps
    res["step"][key + "_dispatches_per_step"] = len(v) / steps
    v = values(run + "_seq", counter, lambda n: True, steps)
    if v:
        res["step_sequential"][key] = sum(v) / steps
        res["step_sequential"][key + "_dispatches_per_step"] = len(v) / steps
    s = values(run + "_seq", counter, lambda n: "trsm_pair_kernel<true>" in n or "trsm_strip8_kernel<true>" in n)
    if s:
        res["strip_kernel"][key] = sum(s) / len(s)
        res["strip_kernel"][key + "_dispatches"] = len(s)
    s = values(run + "_f32", counter, lambda n: "trsm_strip_f32_kernel" in n)
    if s:
        res["f32_strip_kernel"][key] = sum(s) / len(s)
        res["f32_strip_kernel"][key + "_dispatches"] = len(s)
import bench
res["kernel_sources_sha"] = bench.kernel_sources_sha()
res["note"] = (f"bench.py --steps 2 --warmup 1 --post-steps 0 ({steps} steps), N=4096 M=16384; f32: --dtype f32 --steps 1, "
               f"N=16384 M=32768; KB as rocprofv3 reports (FETCH_SIZE is doubled by bench.py, MI355X_MICROARCH.md HBM)")
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
