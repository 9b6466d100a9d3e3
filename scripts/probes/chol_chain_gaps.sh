# The factorisation alone at 4096 points under a kernel trace: what the chain's launch boundaries cost (the upper bound of what a
# persistent chain kernel could take out).  usage: bash scripts/probes/chol_chain_gaps.sh   (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/cholgaps && rm -rf gpurun_out/cholgaps/t
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cholgaps/t -- python3 scripts/chol_timing.py 4096 > gpurun_out/cholgaps/run.log 2>&1
python3 scripts/chol_timeline.py $(ls -t gpurun_out/cholgaps/t/*/*kernel_trace.csv | head -1) > gpurun_out/cholgaps/timeline.txt
python3 - <<'PY'
import re
rows=[l for l in open('gpurun_out/cholgaps/timeline.txt') if ' us ' in l and 'gap' in l]
chain=[l for l in rows if re.search(r'potrf_panel_fused|syrk_rows|potrf_diag128|zero_ints|rhs', l)]
dur=sum(float(re.search(r'\+\s*([0-9.]+) us', l).group(1)) for l in chain)
first=float(rows[0].split()[0]); 
import sys
tot=[l for l in open('gpurun_out/cholgaps/timeline.txt') if l.startswith('total')][0]
# chain-stream gaps: time between the end of one chain kernel and the start of the next chain kernel
ends=[]; gaps=0.0; waits=0.0; prev=None; n=0
for l in chain:
    a=float(l.split()[0]); d=float(re.search(r'\+\s*([0-9.]+) us', l).group(1))
    if prev is not None:
        g=a-prev
        gaps+=g; n+=1
    prev=a+d
print(tot.strip()); print(f"chain kernels: {len(chain)} launches, {dur:.1f} us of kernel time, {gaps:.1f} us between them ({gaps/max(n,1):.2f} us per boundary, waits for the bulk update included)")
PY
tail -3 gpurun_out/cholgaps/run.log
