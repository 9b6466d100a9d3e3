"""The sweep's strip kernel alone (V = L^-1 K*, fused sum V^2 and V^T z) over problem sizes: ms per launch, TFLOP/s,
fraction of the fp64 MFMA peak, and the two-parameter fit  T = a * (regular stages) + b * (row blocks)  that separates
the steady-state stage cost from the per-block (diagonal phase) cost without instrumenting the kernel.
usage: python scripts/strip_scaling.py [M] [n ...]"""
import os, sys
os.environ["CBO_HIP_SWEEP_CACHE"] = "0"          # every sweep runs the kernel
os.environ.setdefault("CBO_HIP_SWEEP", "0")        # left-looking: one launch of the strip kernel per sweep
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib, CausalExpectedImprovement
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.utils_functions.causal_acquisition_functions import CandidateGrid

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sizes = [int(a) for a in sys.argv[2:]] or [1024, 2048, 4096, 8192]
rng = np.random.default_rng(0)
ctx = _lib.Context.get(0)
rows = []
for n in sizes:
    X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
    y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
    Xs = rng.uniform([-5, -5, -5], [5, 20, 5], (M, 3))
    model = HipGaussianProcess(X, y)
    grid = CandidateGrid(Xs, model)
    ei = CausalExpectedImprovement(float(y.min()), "min", model)
    for _ in range(2):
        ei.sweep(grid, cost=3.0)
    ctx.set_profiling(True)
    ctx.reset_timers()
    reps = 5
    for _ in range(reps):
        ei.sweep(grid, cost=3.0)
    t = ctx.timers()
    ctx.set_profiling(False)
    launches = max(1, int(t["n_trsm_launches"]))
    ms = t["ms_trsm"] / launches
    tf = t["trsm_flops"] / launches / ms / 1e9
    blocks = (n + 127) // 128
    rows.append((n, ms, blocks * (blocks - 1) * 2, blocks))
    print(f"n={n:6d} M={M}: {ms:8.4f} ms/launch  {tf:6.1f} TFLOP/s  {tf / 78.6:.3f} of peak  ({launches} launches)")
    grid.close()
if len(rows) >= 2:
    A = np.array([[r[2], r[3]] for r in rows], float)
    b = np.array([r[1] for r in rows]) * 1e3
    (a_, b_), *_ = np.linalg.lstsq(A, b, rcond=None)
    print(f"fit: {a_:.4f} us per regular stage (ideal 64 MFMA x 64 cycles = 1.707 us at 2.4 GHz), {b_:.3f} us per row block")
