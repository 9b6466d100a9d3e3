"""The append-only trial step at BASELINE config 2 size: one new observation, then the sweep of the 16384-candidate
grid -- through cbo_gp_append + one new row of the resident V, against the overlapped refit + sweep (what bench.py
times) and the two calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CandidateGrid, CausalExpectedImprovement, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.graphs import meshgrid_candidates
BOX = [(-5.0, 5.0), (-5.0, 20.0), (-5.0, 5.0)]
lo, hi = np.array([b[0] for b in BOX]), np.array([b[1] for b in BOX])
rng = np.random.default_rng(0)
n0 = 4000
f = lambda X: (np.cos(np.exp(-X[:, 0] / 3)) - np.exp(-X[:, 1] / 20) + 0.3 * np.sin(X[:, 2]))[:, None]
X = rng.uniform(lo, hi, (n0, 3)); y = f(X) + 0.1 * rng.standard_normal((n0, 1))
Xs = meshgrid_candidates(BOX, [32, 32, 16])
ctx = _lib.Context.get()
m = HipGaussianProcess(X, y)
grid = CandidateGrid(Xs, m, keep_solution=True)
ei = lambda: CausalExpectedImprovement(float(y.min()), "min", m)
ei().sweep(grid)
t_app, t_sweep = [], []
for _ in range(40):
    x_new = rng.uniform(lo, hi, (1, 3)); y_new = f(x_new) + 0.1 * rng.standard_normal((1, 1))
    t0 = time.perf_counter(); ok = m.append(x_new, y_new); t1 = time.perf_counter()
    assert ok
    X = np.vstack([X, x_new]); y = np.vstack([y, y_new])
    r = ei().sweep(grid); t2 = time.perf_counter()
    t_app.append(t1 - t0); t_sweep.append(t2 - t1)
print(f"append-only step at N={X.shape[0]}, M={Xs.shape[0]}: append {np.median(t_app)*1e3:.2f} ms + sweep (one new row) "
      f"{np.median(t_sweep)*1e3:.2f} ms = {np.median(np.add(t_app, t_sweep))*1e3:.2f} ms per trial")
ref = HipGaussianProcess(X, y)
full = CausalExpectedImprovement(float(y.min()), "min", ref).sweep(Xs, want_acq=True)
inc = ei().sweep(grid, want_acq=True)
sys.stdout.flush()
print("same winner as a model fitted from scratch:", full["best_idx"] == inc["best_idx"],
      "max |acq diff| / max acq:", float(np.max(np.abs(full["acq"] - inc["acq"])) / full["acq"].max()))
t0 = time.perf_counter()
for _ in range(5):
    ref.set_data(X, y, fit=False); CausalExpectedImprovement(float(y.min()), "min", ref).sweep(Xs)
print(f"rebuild: upload + overlapped refit + sweep {(time.perf_counter()-t0)/5*1e3:.2f} ms per trial")
