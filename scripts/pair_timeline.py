"""Timing-only (diagnostic build): per-stage s_memtime stamps of workgroup 0, waves 0 (block b of every pair) and 4 (block
b + 1), of trsm_pair_kernel.  Run with CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so.  Prints, for the pairs
from row 1024 on, the mean period of the regular stages by position and of every diagonal stage, with where each role
passed the stage's milestones.  usage: python scripts/pair_timeline.py [n]"""
import ctypes, os, sys
os.environ.setdefault("CBO_HIP_STRIP_MASK", "256")
os.environ["CBO_HIP_SWEEP_CACHE"] = "0"
os.environ.setdefault("CBO_HIP_SWEEP", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib, CausalExpectedImprovement
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.graphs import meshgrid_candidates
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
Xs = meshgrid_candidates([(-5, 5), (-5, 20), (-5, 5)], (32, 32, 16))
m = HipGaussianProcess(X, y)
ei = CausalExpectedImprovement(float(y.min()), "min", m)
for _ in range(3):
    ei.sweep(Xs, cost=3.0)
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 4096))()
lib.cbo_diag_trsm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cbo_diag_trsm_stamps(buf, 8 * 4096) == 0
kinds = []
for i0 in range(0, n, 256):
    kinds += [("reg", j, i0) for j in range(i0 // 16)] + [("d0", m_, i0) for m_ in range(8)] + [("d1", m_, i0) for m_ in range(4)]
nst = min(len(kinds), 4096)
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:nst].astype(np.int64)
kinds = kinds[:nst]
top = st[:, 0]                                    # wave 0 passes the stage top
per = np.append(top[1:] - top[:-1], 0)
print(f"n = {n}: {nst} stages stamped, {top[-1] - top[0]} cycles from the first top to the last")
sel = lambda f: np.array([f(k) for k in kinds])
big = sel(lambda k: k[2] >= 1024)
reg = sel(lambda k: k[0] == "reg")
print("regular stages (pairs from row 1024): mean period", per[big & reg].mean().round(0), " by position 0..5:",
      [int(per[big & reg & sel(lambda k: k[1] == j)].mean()) for j in range(6)], " last three of a pair:",
      [int(per[big & reg & sel(lambda k: k[1] == k[2] // 16 - 1 - j)].mean()) for j in (2, 1, 0)])
for kind, cnt in (("d0", 8), ("d1", 4)):
    for m_ in range(cnt):
        s = big & sel(lambda k: k[0] == kind and k[1] == m_)
        r = st[s]
        arr = f" [at the mid barrier after {np.mean(r[:, 3] - r[:, 0]):5.0f} / {np.mean(r[:, 7] - r[:, 4]):5.0f}]" if kind == "d0" else ""
        w0 = f"top->mid {np.mean(r[:, 1] - r[:, 0]):6.0f} mid->end {np.mean(r[:, 2] - r[:, 1]):6.0f}" if kind == "d0" else f"top->end {np.mean(r[:, 2] - r[:, 0]):6.0f}"
        print(f"  {kind} stage {m_}: period {per[s].mean():7.0f}   wave 0: {w0}"
              f"   wave 4: top (after wave 0's) {np.mean(r[:, 4] - r[:, 0]):5.0f} top->mid {np.mean(r[:, 5] - r[:, 4]):6.0f} mid->end {np.mean(r[:, 6] - r[:, 5]):6.0f}{arr}")
tot = lambda kk: per[big & sel(lambda k: k[0] == kk)].sum() / max(1, len(set(k[2] for k in kinds if k[2] >= 1024)))
print(f"per pair (from row 1024): D0 {tot('d0'):.0f} cycles, D1 {tot('d1'):.0f} cycles")
