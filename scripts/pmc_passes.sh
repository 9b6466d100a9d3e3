#!/bin/bash
# PMC counter passes for the dominant kernels (run on the GPU box from the repo root).
# Each --pmc set is its own rocprofv3 run (gfx950: 8 SQ slots, FETCH_SIZE=3 + WRITE_SIZE=2 of 4 TCC slots); the
# program comes directly after `--` (no wrapper).  Then scripts/pmc_to_json.py OUT profiles/pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
MODE=""
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --post-steps 0 $MODE > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
# The overlapped step's schedule is FORCED to the one the context settles on at this shape outside the profiler (4 of 16
# pairs pipelined, updates in groups of two: profiles/r04_bench_default.json config.schedule): counter collection
# serialises the kernels, the overlap gains nothing there, and a context left to measure would settle on the plain sequence.
# PMC_ONLY=overlapped runs just these three passes.
export CBO_HIP_OVERLAP=1 CBO_HIP_PIPE_TAIL=0.75 CBO_HIP_PIPE_GROUP=2
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE
unset CBO_HIP_OVERLAP CBO_HIP_PIPE_TAIL CBO_HIP_PIPE_GROUP
if [ "$PMC_ONLY" = "overlapped" ]; then python3 scripts/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1; exit 0; fi
MODE="--sequential" &&
run fetch_seq FETCH_SIZE &&
run write_seq WRITE_SIZE &&
run sq_seq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE &&
run sq2_seq SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM &&
MODE="--config c5 --steps 1" &&
run fetch_f32 FETCH_SIZE &&
run write_f32 WRITE_SIZE &&
run sq_f32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE &&
run sq2_f32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
python3 scripts/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
