"""The coral graph's count of exploration sets (S = 25) with this package's default grids (200 / 64x64 / 32x32x16 candidates for
d = 1 / 2 / 3) and 50 observations per set: latency of the one multi-set call (CBO_HIP_SMALL_TWO_PHASE=0: every workgroup
factors its set's model itself; default: one workgroup per set factors first when a set has 12 or more candidate blocks)."""
import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.graphs import meshgrid_candidates
rng = np.random.default_rng(0)
lib = _lib.load()
S, n = 25, 50
models, grids = [], []
for s in range(S):
    d = 1 + s % 3
    box = [(-1.0, 2.0)] * d
    X = rng.uniform(-1, 2, (n, d)); y = np.cos(X).sum(1, keepdims=True)
    models.append(HipGaussianProcess(X, y, noise_var=1e-3, fit=False))
    shape = [(200,), (64, 64), (32, 32, 16)][d - 1]
    grids.append(CandidateGrid(meshgrid_candidates(box, shape), models[-1]))
gps = (ctypes.c_void_p * S)(*[m._handle for m in models]); cds = (ctypes.c_void_p * S)(*[g._handle for g in grids])
yb, cs, vals, idxs = np.full(S, 0.1), np.ones(S), np.empty(S), np.empty(S, dtype=np.int64)
args = (S, gps, cds, _lib.dptr(yb), 0, 0.0, _lib.dptr(cs), _lib.dptr(vals), idxs.ctypes.data_as(_lib.c_int64_p))
for _ in range(5): _lib.check(lib.cbo_acq_sweep_sets(*args))
ts = []
for _ in range(200):
    t0 = time.perf_counter(); lib.cbo_acq_sweep_sets(*args); ts.append(time.perf_counter() - t0)
print(f"25 sets (d = 1, 2, 3: 200 / 4096 / 16384 candidates), 50 observations each: median {np.median(ts)*1e6:.0f} us per call; checksum {vals.sum():.12e} {idxs.sum()}")
