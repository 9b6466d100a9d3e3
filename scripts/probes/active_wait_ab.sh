cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05w
for rep in 1 2 3; do
  for v in 0 20000; do
    ROC_ACTIVE_WAIT_TIMEOUT=$v timeout -k 10 120 python3 bench.py --steps 40 --warmup 3 --cpu-sample 0 --post-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ROC_ACTIVE_WAIT_TIMEOUT=$v', round(d['ms_per_step'],4))"
  done
done
