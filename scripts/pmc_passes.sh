#!/bin/bash
# PMC counter passes for the dominant kernels (run on the GPU box from the repo root).
# Each --pmc set is its own rocprofv3 run (gfx950: 8 SQ slots, FETCH_SIZE=3 + WRITE_SIZE=2 of 4 TCC slots).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
MODE=""
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --post-steps 0 $MODE > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum &&
MODE="--sequential" &&
run fetch_seq FETCH_SIZE &&
run write_seq WRITE_SIZE &&
run sq_seq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
