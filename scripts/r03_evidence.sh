#!/bin/bash
# Round-3 evidence run on the GPU box (from the repo root): bench lines for every BASELINE config, rocprofv3 kernel
# stats of the same commands, auxiliary timings.  Outputs under gpurun_out/r03/ ; the summaries worth keeping are copied
# to profiles/ afterwards (scripts/pmc_passes.sh is a separate call: one counter set per rocprofv3 run).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python3 bench.py --sequential --cpu-sample 0 > $OUT/bench_sequential.json 2> $OUT/bench_sequential.err; echo "bench sequential rc=$?"
for c in c1 c3 c4 c5; do
  timeout -k 10 500 python3 bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 > $OUT/prof_default.log 2>&1; echo "prof default rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sequential -- python3 bench.py --sequential --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 > $OUT/prof_sequential.log 2>&1; echo "prof sequential rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -- python3 bench.py --config c5 --steps 4 --warmup 1 --cpu-sample 0 --post-steps 0 > $OUT/prof_f32.log 2>&1; echo "prof f32 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python3 bench.py --config c3 --steps 4 --warmup 1 --cpu-sample 0 --post-steps 0 > $OUT/prof_c3.log 2>&1; echo "prof c3 rc=$?"
timeout -k 10 200 python3 scripts/chol_timing.py 1024 2048 4096 8192 16384 > $OUT/chol_timing.txt 2>&1; echo "chol rc=$?"
timeout -k 10 200 python3 scripts/strip_scaling.py 16384 1024 2048 4096 8192 > $OUT/strip_scaling.txt 2>&1; echo "strip rc=$?"
CBO_HIP_STRIP_FORM=4 timeout -k 10 200 python3 scripts/strip_scaling.py 16384 1024 2048 4096 8192 > $OUT/strip_scaling_one_wave.txt 2>&1; echo "strip4 rc=$?"
timeout -k 10 200 python3 scripts/vec_solve_timing.py > $OUT/vec_solve_timing.txt 2>&1; CBO_HIP_VEC_SOLVE_FORM=1 timeout -k 10 200 python3 scripts/vec_solve_timing.py >> $OUT/vec_solve_timing.txt 2>&1; echo "vec rc=$?"
timeout -k 10 200 python3 scripts/append_step_timing.py > $OUT/append_step_timing.txt 2>&1; echo "append rc=$?"
timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 > $OUT/loop.txt 2>&1; timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 --optimize >> $OUT/loop.txt 2>&1; echo "loop rc=$?"
timeout -k 10 120 ./scripts/probes/stage_probe > $OUT/stage_probe.txt 2>&1; echo "probe rc=$?"
# second half of the round: the schedules with grouped updates, the 16384-point factorisation pair by pair and in groups
timeout -k 10 900 python3 scripts/schedule_scan.py > $OUT/schedule_crossover.txt 2>&1; echo "schedule scan rc=$?"
for g in 1 2; do
  rm -rf $OUT/tl16_g$g
  CBO_HIP_BULK_GROUP=$g timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl16_g$g -- python3 scripts/chol_timing.py 16384 > $OUT/tl16_g$g.log 2>&1; echo "chol trace group=$g rc=$?"
  python3 scripts/chol_timeline.py $(ls -t $OUT/tl16_g$g/*/*kernel_trace.csv | head -1) > $OUT/chol_timeline_16384_group$g.txt
done
CBO_HIP_BULK_GROUP=1 timeout -k 10 200 python3 scripts/chol_timing.py 8192 16384 > $OUT/chol_timing_pairs.txt 2>&1; echo "chol pairs rc=$?"
CBO_HIP_PIPE_GROUP=1 CBO_HIP_PIPE_CHUNK=2 timeout -k 10 300 python3 bench.py --cpu-sample 0 --post-steps 0 > $OUT/bench_default_pairs.json 2> $OUT/bench_default_pairs.err; echo "bench pairs rc=$?"
rm -rf $OUT/step_trace; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/step_trace -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --post-steps 0 > $OUT/step_trace.log 2>&1; echo "step trace rc=$?"
python3 scripts/step_timeline.py $(ls -t $OUT/step_trace/*/*kernel_trace.csv | head -1) > $OUT/step_timeline.txt
CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$GRAFT_REPO_ROOT/cbo_with_oop_amd/libcbo_hip_diag.so timeout -k 10 300 python3 scripts/update_kernel_timing.py 16384 8192 > $OUT/update_kernel_timing.txt 2>&1; echo "update kernel rc=$?"
for d in default sequential f32 c3; do cp $OUT/prof_$d/*/*kernel_stats.csv $OUT/kernel_stats_$d.csv; done
