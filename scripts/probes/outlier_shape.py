"""The schedule scan's outlier shape (2048 observations x 16384 candidates), one PROCESS per schedule: forced splits and the
settled automatic one.  usage: python scripts/probes/outlier_shape.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, %r)
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
n, mm, d = 2048, 16384, 3
ctx = _lib.Context.get(0); lib = _lib.load()
rng = np.random.default_rng(n + mm + d)
X = rng.uniform(-5, 5, (n, d)); y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
Xs = rng.uniform(-5, 5, (mm, d))
m = HipGaussianProcess(X, y, noise_var=1e-2, fit=False, context=ctx); g = CandidateGrid(Xs, m, context=ctx)
bv, bi = ctypes.c_double(), ctypes.c_int64()
call = lambda: _lib.check(lib.cbo_gp_fit_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None, ctypes.byref(bv), ctypes.byref(bi), None, None))
call(); call()
k = 0
while ctx.schedule_report()[0] > 0 and k < 80: call(); k += 1
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10): call()
    best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
print("%%.3f ms  %%s" %% (best, ctx.schedule_report()[1].strip().splitlines()[0][:150] if k else ""))
""" % ROOT
for name, env in [("auto", {}), ("seq", {"CBO_HIP_OVERLAP": "0"})] + [(f"g{g}/{t}", {"CBO_HIP_OVERLAP": "1", "CBO_HIP_PIPE_TAIL": str(t), "CBO_HIP_PIPE_GROUP": g}) for g in ("1", "2") for t in (1.0, 0.875, 0.75, 0.625, 0.5)]:
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    print(f"{name:10s} {r.stdout.strip() if r.returncode == 0 else r.stderr[-300:]}", flush=True)
