#!/bin/bash
# Timing-only: price the phases of the Cholesky kernels with the diagnostic build (results are wrong by design).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$GRAFT_REPO_ROOT/cbo_with_oop_amd/libcbo_hip_diag.so
for mask in ${MASKS:-0 1 2 4 8 16 32 63 64 128 192 256}; do
  export CBO_DBG_CHOL=$mask
  rm -rf gpurun_out/diag_$mask
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/diag_$mask -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/diag_$mask.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/diag_$mask/*/*kernel_stats.csv")
rows={r["Name"][:28]:r for r in csv.DictReader(open(f[0]))} if f else {}
def g(k):
    for n,r in rows.items():
        if k in n: return f'{float(r["AverageNs"])/1e3:7.1f}us(max {float(r["MaxNs"])/1e3:6.1f})'
    return "   -"
print(f"mask=$mask diag={g('potrf_diag128')} syrk={g('syrk_kernel')} panel={g('trsm_strip_kernel<false')}")
PY
done
