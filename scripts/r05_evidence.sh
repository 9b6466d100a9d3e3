#!/bin/bash
# Round-5 evidence run on the GPU box (from the repo root), on the final sources: bench lines for every BASELINE config,
# rocprofv3 kernel stats of the same commands, the schedule scan, the tolerance report and the auxiliary timings.  Outputs
# under gpurun_out/r05/ ; the summaries worth keeping are copied to profiles/ afterwards (scripts/pmc_passes.sh is a
# separate call: one counter set per rocprofv3 run).  usage: scripts/r05_evidence.sh a|b|c|d (one gpurun call each: a = bench lines and kernel stats, b = strong-scaling model and schedule scan, d = everything else, c = the subset of d a change of the factorisation touches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05
mkdir -p $OUT
if [ "$1" = "a" ]; then
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout -k 10 300 python3 bench.py --sequential --cpu-sample 0 > $OUT/bench_sequential.json 2> $OUT/bench_sequential.err; echo "bench sequential rc=$?"
for c in c1 c3 c4 c5; do
  timeout -k 10 500 python3 bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "bench $c rc=$?"
done
CBO_BENCH_SELF_LAUNCH=1 timeout -k 10 300 python3 bench.py --gpus 1 --cpu-sample 0 --post-steps 0 > $OUT/bench_self_launch.json 2> $OUT/bench_self_launch.err; echo "bench self-launch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 > $OUT/prof_default.log 2>&1; echo "prof default rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sequential -- python3 bench.py --sequential --steps 20 --warmup 3 --cpu-sample 0 --post-steps 0 > $OUT/prof_sequential.log 2>&1; echo "prof sequential rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c1 -- python3 bench.py --config c1 --cpu-sample 0 > $OUT/prof_c1.log 2>&1; echo "prof c1 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -- python3 bench.py --config c5 --steps 4 --warmup 1 --cpu-sample 0 --post-steps 0 > $OUT/prof_f32.log 2>&1; echo "prof f32 rc=$?"
rm -rf $OUT/step_trace; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/step_trace -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --post-steps 0 > $OUT/step_trace.log 2>&1; echo "step trace rc=$?"
python3 scripts/step_timeline.py $(ls -t $OUT/step_trace/*/*kernel_trace.csv | head -1) > $OUT/step_timeline.txt
for d in default sequential c1 f32; do cp $OUT/prof_$d/*/*kernel_stats.csv $OUT/kernel_stats_$d.csv; done
elif [ "$1" = "c" ]; then
# (the pieces a change of the factorisation's build touches)
timeout -k 10 200 python3 scripts/chol_timing.py 1024 2048 4096 8192 16384 > $OUT/chol_timing.txt 2>&1; echo "chol rc=$?"
timeout -k 10 200 python3 scripts/probes/trial_step_breakdown.py > $OUT/trial_step_breakdown.txt 2>&1; echo "trial step rc=$?"
CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$GRAFT_REPO_ROOT/cbo_with_oop_amd/libcbo_hip_diag.so timeout -k 10 200 python3 scripts/small_stamps.py > $OUT/small_stamps.txt 2>&1; echo "small stamps rc=$?"
CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$GRAFT_REPO_ROOT/cbo_with_oop_amd/libcbo_hip_diag.so timeout -k 10 200 python3 scripts/diag_stamps.py 128 > $OUT/diag_stamps.txt 2>&1; echo "diag stamps rc=$?"
timeout -k 10 100 ./scripts/probes/tile_factor_probe > $OUT/tile_factor_probe.txt 2>&1; echo "probe rc=$?"
timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 > $OUT/loop.txt 2>&1; timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 --optimize >> $OUT/loop.txt 2>&1; echo "loop rc=$?"
timeout -k 10 200 python3 scripts/small_config_latency.py > $OUT/small_config_latency.txt 2>&1; echo "small config rc=$?"
elif [ "$1" = "b" ]; then
timeout -k 10 600 python3 scripts/strong_model.py c2 c3 > $OUT/strong_model.txt 2>&1; echo "strong model rc=$?"
timeout -k 10 900 python3 scripts/schedule_scan.py > $OUT/schedule_crossover.txt 2> $OUT/schedule_crossover.err; echo "schedule scan rc=$?"
else
timeout -k 10 600 python3 scripts/tolerance_report.py --large > $OUT/tolerance_report.txt 2> $OUT/tolerance_report.err; echo "tolerance rc=$?"
timeout -k 10 200 python3 scripts/chol_timing.py 1024 2048 4096 8192 16384 > $OUT/chol_timing.txt 2>&1; echo "chol rc=$?"
timeout -k 10 200 python3 scripts/strip_scaling.py 16384 1024 2048 4096 8192 > $OUT/strip_scaling.txt 2>&1; echo "strip rc=$?"
CBO_HIP_STRIP_FORM=8 timeout -k 10 200 python3 scripts/strip_scaling.py 16384 1024 2048 4096 8192 > $OUT/strip_scaling_strip8.txt 2>&1; echo "strip8 rc=$?"
timeout -k 10 300 python3 scripts/strip_form_bits.py > $OUT/strip_form_bits.txt 2>&1; echo "bits rc=$?"
CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$GRAFT_REPO_ROOT/cbo_with_oop_amd/libcbo_hip_diag.so timeout -k 10 200 python3 scripts/pair_timeline.py > $OUT/pair_timeline.txt 2>&1; echo "pair timeline rc=$?"
timeout -k 10 200 python3 scripts/ei_pass_timing.py 20 22 24 > $OUT/ei_pass_timing.txt 2>&1; echo "ei rc=$?"
timeout -k 10 400 python3 scripts/lib_ab.py cbo_with_oop_amd/libcbo_hip.so@STRIP_FORM=8 cbo_with_oop_amd/libcbo_hip.so > $OUT/lib_ab.txt 2>&1; echo "ab rc=$?"
timeout -k 10 200 python3 scripts/probes/trial_step_breakdown.py > $OUT/trial_step_breakdown.txt 2>&1; echo "trial step rc=$?"
CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$GRAFT_REPO_ROOT/cbo_with_oop_amd/libcbo_hip_diag.so timeout -k 10 200 python3 scripts/small_stamps.py > $OUT/small_stamps.txt 2>&1; echo "small stamps rc=$?"
timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 > $OUT/loop.txt 2>&1; timeout -k 10 300 python3 scripts/complete_graph_cbo_loop.py 20 --optimize >> $OUT/loop.txt 2>&1; echo "loop rc=$?"
timeout -k 10 200 python3 scripts/small_config_latency.py > $OUT/small_config_latency.txt 2>&1; echo "small config rc=$?"
fi
