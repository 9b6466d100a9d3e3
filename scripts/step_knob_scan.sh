#!/bin/bash
# The overlapped C2 step (bench.py default) under combinations of environment knobs, one line per combination:
# ms_per_step of 20 timed steps.  Usage on the GPU box, from the repo root:  scripts/step_knob_scan.sh "K1=v K2=v" "K3=v" ...
# (an empty string = the defaults).  Output: gpurun_out/knob_scan.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/knob_scan.txt
: > $OUT
for combo in "$@"; do
  line=$(env $combo timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 2> gpurun_out/knob_scan.err)
  rc=$?
  ms=$(echo "$line" | python3 -c "import sys, json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "rc=$rc ms_per_step=$ms  [$combo]" | tee -a $OUT
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/knob_scan.err | tee -a $OUT; break; fi
done
