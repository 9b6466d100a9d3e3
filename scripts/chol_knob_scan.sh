#!/bin/bash
# scripts/chol_timing.py under combinations of environment knobs.  Usage on the GPU box, from the repo root:
#   scripts/chol_knob_scan.sh "<sizes>" "K1=v K2=v" "K3=v" ...     Output: gpurun_out/chol_knob_scan.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/chol_knob_scan.txt
: > $OUT
sizes=$1; shift
for combo in "$@"; do
  echo "[$combo]" | tee -a $OUT
  env $combo timeout -k 10 200 python3 scripts/chol_timing.py $sizes 2>&1 | tee -a $OUT
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "rc=$rc" | tee -a $OUT; break; fi
done
