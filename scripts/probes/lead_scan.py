"""The pipelined sweep with LEAD pairs going alone ahead of the first group (CBO_HIP_PIPE_LEAD): step time at BASELINE config 2 and
a digest of the sweep's outputs (same bits or not), one process per schedule.  usage: python scripts/probes/lead_scan.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import ctypes, hashlib, sys, time
import numpy as np
sys.path.insert(0, %r)
import bench
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
cfg = bench.CONFIGS["c2"]
X, y, Xs, grid, note = bench.make_problem(cfg, 1, "weak", False)
ctx = _lib.Context.get(0); lib = _lib.load()
m = HipGaussianProcess(X, y, context=ctx, fit=False); g = CandidateGrid(Xs, m, context=ctx)
bv, bi = ctypes.c_double(), ctypes.c_int64()
n = Xs.shape[0]
acq, mean, var = np.empty(n), np.empty(n), np.empty(n)
def call(outs=False):
    _lib.check(lib.cbo_gp_fit_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, _lib.dptr(acq) if outs else None,
               _lib.dptr(mean) if outs else None, _lib.dptr(var) if outs else None, ctypes.byref(bv), ctypes.byref(bi), None, None))
for _ in range(12): call()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10): call()
    best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
call(True)
h = hashlib.sha256(); [h.update(np.ascontiguousarray(a).tobytes()) for a in (acq, mean, var)]
print("%%.3f ms  digest %%s  winner %%d" %% (best, h.hexdigest()[:12], bi.value))
""" % ROOT
combos = [("0.75", "2", "0"), ("0.8125", "2", "1"), ("0.75", "2", "2"), ("0.75", "3", "1"), ("0.8125", "1", "0"), ("0.875", "2", "0"),
          ("0.8125", "2", "0"), ("0.6875", "2", "1"), ("0.6875", "4", "1"), ("0.75", "1", "0")]
if len(sys.argv) > 1:
    combos = [tuple(a.split(",")) for a in sys.argv[1:]]
for tail, grp, lead in combos:
    env = dict(os.environ, CBO_HIP_OVERLAP="1", CBO_HIP_PIPE_TAIL=tail, CBO_HIP_PIPE_GROUP=grp, CBO_HIP_PIPE_LEAD=lead)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    print(f"pairs {round((1 - float(tail)) * 16)} groups of {grp} lead {lead}: {r.stdout.strip() if r.returncode == 0 else r.stderr[-400:]}", flush=True)
