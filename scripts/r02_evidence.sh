#!/bin/bash
# Round-2 evidence run on the GPU box (from the repo root): bench lines, rocprofv3 kernel stats of the same commands,
# auxiliary timings.  Outputs under gpurun_out/r02/ ; the summaries worth keeping are copied to profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02
mkdir -p $OUT
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python3 bench.py --sequential --cpu-sample 0 > $OUT/bench_sequential.json 2> $OUT/bench_sequential.err; echo "bench sequential rc=$?"
timeout -k 10 600 python3 bench.py --dtype f32 > $OUT/bench_f32.json 2> $OUT/bench_f32.err; echo "bench f32 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 > $OUT/prof_default.log 2>&1; echo "prof default rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sequential -- python3 bench.py --sequential --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 > $OUT/prof_sequential.log 2>&1; echo "prof sequential rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -- python3 bench.py --dtype f32 --steps 4 --warmup 1 --cpu-sample 0 --post-steps 0 > $OUT/prof_f32.log 2>&1; echo "prof f32 rc=$?"
timeout -k 10 200 python3 scripts/chol_timing.py 1024 2048 4096 8192 16384 > $OUT/chol_timing.txt 2>&1; echo "chol rc=$?"
timeout -k 10 200 python3 scripts/small_config_latency.py > $OUT/small_config_latency.txt 2>&1; echo "small rc=$?"
timeout -k 10 200 python3 scripts/f32_check.py 500 2048 4096 > $OUT/f32_check.txt 2>&1; echo "f32 check rc=$?"
timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 > $OUT/loop.txt 2>&1; timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 --optimize >> $OUT/loop.txt 2>&1; echo "loop rc=$?"
for d in default sequential f32; do cp $OUT/prof_$d/*/*kernel_stats.csv $OUT/kernel_stats_$d.csv; done
