"""HBM traffic from the rocprofv3 --pmc passes (scripts/pmc_passes.sh output) into profiles/pmc_traffic.json,
which bench.py reads for roofline.traffic:
  "step"             : FETCH_SIZE / WRITE_SIZE (KB) summed over every kernel dispatch of the run, divided by the
                       number of steps (bench.py --steps 2 --warmup 1 --post-steps 0: three overlapped steps; the model
                       is created unfitted, so nothing else launches kernels)
  "step_sequential"  : the same for --sequential
  "strip_kernel"     : mean per dispatch of trsm_strip8_kernel<true> (round 2: trsm_strip_kernel<true, 32>) from the
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
def values(run, counter, pred):
    out = []
    for f in glob.glob(f"{root}/{run}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"].startswith(counter) and pred(row["Kernel_Name"]):
                out.append(float(row["Counter_Value"]))
    return out
res = {"step": {}, "step_sequential": {}, "strip_kernel": {}, "f32_strip_kernel": {}}
for run, counter, key in (("fetch", "FETCH_SIZE", "fetch_size_kb"), ("write", "WRITE_SIZE", "write_size_kb")):
    v = values(run, counter, lambda n: True)
    res["step"][key] = sum(v) / steps
    res["step"][key + "_dispatches_per_step"] = len(v) / steps
    v = values(run + "_seq", counter, lambda n: True)
    if v:
        res["step_sequential"][key] = sum(v) / steps
        res["step_sequential"][key + "_dispatches_per_step"] = len(v) / steps
    s = values(run + "_seq", counter, lambda n: "trsm_strip8_kernel<true>" in n or "trsm_strip_kernel<true, 32>" in n)
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
