"""A/B of several BUILDS of libcbo_hip.so on one box, interleaved, one process per measurement: the plain left-looking
sweep's strip kernel (per-launch hipEvents), the fit, and the overlapped refit + sweep step, at the C2 shape.  Box-to-box
spread on this pool is 3-5 %, so only same-box comparisons count.  A library is named  path.so[@KNOB=value...]  (KNOB without
the CBO_HIP_ prefix, set in that measurement's environment).
usage: python scripts/lib_ab.py lib1.so lib2.so@PIPE_GROUP=1 ... [-- n m]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, %r)
import bench
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
n, m = int(sys.argv[1]), int(sys.argv[2])
cfg = bench.CONFIGS["c2"]
X, y, Xs, grid, note = bench.make_problem(cfg, 1, "weak", False)
X, y, Xs = X[:n], y[:n], Xs[:m]
ctx = _lib.Context.get(0)
lib = _lib.load()
model = HipGaussianProcess(X, y, context=ctx)
cands = CandidateGrid(Xs, model, context=ctx)
bv, bi = ctypes.c_double(), ctypes.c_int64()
acq, mean, var = np.empty(m), np.empty(m), np.empty(m)
def sweep(outs=False):
    _lib.check(lib.cbo_gp_fit(model._handle, None, None))
    _lib.check(lib.cbo_acq_sweep(model._handle, cands._handle, float(y.min()), 0, 0.0, 3.0, _lib.dptr(acq) if outs else None,
                                 _lib.dptr(mean) if outs else None, _lib.dptr(var) if outs else None, ctypes.byref(bv), ctypes.byref(bi)))
def fused():
    _lib.check(lib.cbo_gp_fit_sweep(model._handle, cands._handle, float(y.min()), 0, 0.0, 3.0, None, None, None,
                                    ctypes.byref(bv), ctypes.byref(bi), None, None))
for _ in range(3): sweep()
ctx.set_profiling(True); ctx.reset_timers()
for _ in range(10): sweep()
t = ctx.timers(); ctx.set_profiling(False)
for _ in range(3): fused()
settle = 0
while ctx.schedule_report()[0] > 0 and settle < 90:       # cbo_gp_fit_sweep measures its way to a schedule first
    fused(); settle += 1
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(20): fused()
ctx.synchronize(); step = (time.perf_counter() - t0) / 20
sweep(True)
import hashlib
print(json.dumps({"trsm_ms": t["ms_trsm"] / t["n_trsm_launches"], "chol_ms": t["ms_chol"] / t["n_fit"], "step_ms": step * 1e3,
                  "best": [bv.value, bi.value], "var": var.tolist()[:4096:37], "mean": mean.tolist()[:4096:37],
                  "acq": acq.tolist()[:4096:37]}))
""" % ROOT


args = sys.argv[1:]
n, m = 4096, 16384
if "--" in args:
    k = args.index("--"); n, m = int(args[k + 1]), int(args[k + 2]); args = args[:k]
rows = {a: [] for a in args}
for rep in range(3):
    for lib in args:
        env = dict(os.environ)
        spec = lib.split("@")                       # lib.so@KAPPA=0
        env["CBO_HIP_LIB"] = os.path.abspath(spec[0])
        for kv in spec[1:]:
            k, v = kv.split("=")
            env["CBO_HIP_" + k] = v
        out = subprocess.run([sys.executable, "-c", CODE, str(n), str(m)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        rows[lib].append((d["trsm_ms"], d["chol_ms"], d["step_ms"]))
for lib, r in rows.items():
    print(f"{lib:60s} strip kernel {min(x[0] for x in r):.3f} ms (runs {[round(x[0], 3) for x in r]})  fit {min(x[1] for x in r):.3f}  overlapped step {min(x[2] for x in r):.3f} (runs {[round(x[2], 3) for x in r]})")
