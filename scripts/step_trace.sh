#!/bin/bash
# rocprofv3 --kernel-trace of a short bench.py run under the given environment knobs; the trace lands in
# gpurun_out/<name>/ (scripts/step_timeline.py reads it).  Usage: scripts/step_trace.sh <name> [bench args...]   (knobs
# exported by the caller).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; shift
rm -rf gpurun_out/$name
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$name -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --post-steps 0 "$@" > gpurun_out/$name.log 2>&1
echo "trace $name rc=$?"
