"""Extract FETCH_SIZE / WRITE_SIZE (KB, mean per dispatch) of the strip TRSM from the rocprofv3 --pmc passes
(scripts/pmc_passes.sh output) into profiles/trsm_pmc.json, which bench.py reads for roofline.traffic."""
import csv, glob, json, sys
root = sys.argv[1]
out = {}
for name, key in (("fetch", "fetch_size_kb"), ("write", "write_size_kb")):
    vals = []
    for f in glob.glob(f"{root}/{name}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "trsm_strip_kernel<true>" in row["Kernel_Name"] and row["Counter_Name"].startswith(name.upper()):
                vals.append(float(row["Counter_Value"]))
    out[key] = sum(vals) / len(vals)
    out[key + "_dispatches"] = len(vals)
out["note"] = "mean per dispatch of trsm_strip_kernel<true>, bench.py --steps 2 --warmup 1, N=4096 M=16384"
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
