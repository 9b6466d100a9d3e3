"""BASELINE config 4 shape per GPU (N=16384 observations, 32768 candidates, d=3): refit + sweep as one overlapped
call against the two calls."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.graphs import CoralGraph, meshgrid_candidates
box = CoralGraph.bounds(["N", "O", "C"])     # (the ranges this test has used since round 1)
lo, hi = np.array([b[0] for b in box], float), np.array([b[1] for b in box], float)
rng = np.random.default_rng(16384)
X = rng.uniform(lo, hi, (16384, 3))
y = (np.sin(X[:, 0]) + np.cos(3 * X[:, 1]) * X[:, 2] + 0.1 * rng.standard_normal(16384))[:, None]
Xs = meshgrid_candidates(box, [64, 64, 64])[:32768]
lib, ctx = _lib.load(), _lib.Context.get()
m = HipGaussianProcess(X, y, fit=False)
g = CandidateGrid(Xs, m)
bv, bi = ctypes.c_double(), ctypes.c_int64()
def fused():
    _lib.check(lib.cbo_gp_fit_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None, ctypes.byref(bv), ctypes.byref(bi), None, None))
def two():
    _lib.check(lib.cbo_gp_fit(m._handle, None, None))
    _lib.check(lib.cbo_acq_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None, ctypes.byref(bv), ctypes.byref(bi)))
flops = 16384.0 ** 2 * 32768 + 16384.0 ** 3 / 3
for name, fn in (("overlapped", fused), ("two calls", two)):
    fn(); ctx.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.1f} ms per step -> {32768/dt/1e3:.1f} k acq/s, {flops/dt/1e12:.1f} TFLOP/s whole step, winner {bi.value}", flush=True)
