"""Timing-only (diagnostic build): per-stage s_memtime stamps of workgroup 0 / wave 0 of the sweep's strip TRSM.
Run with CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so.  Prints, for regular and diagonal stages, the mean cycles spent
(a) waiting at the top (s_waitcnt + barrier) and (b) in the stage body."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib, CausalExpectedImprovement
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
n, grid = 4096, (32, 32, 16)
rng = np.random.default_rng(0)
X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
from cbo_with_oop_amd.graphs import meshgrid_candidates
Xs = meshgrid_candidates([(-5, 5), (-5, 20), (-5, 5)], grid)
m = HipGaussianProcess(X, y)
ei = CausalExpectedImprovement(float(y.min()), "min", m)
for _ in range(3):
    ei.sweep(Xs, cost=3.0)
lib = _lib.load()
nst = sum(i0 // 32 + 4 for i0 in range(0, n, 128))
buf = (ctypes.c_ulonglong * (8 * 4096))()
rc = lib.cbo_diag_trsm_stamps(buf, 8 * 4096)
st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:min(nst, 4096)].astype(np.int64)
kinds = []
for i0 in range(0, n, 128):
    kinds += ["reg"] * (i0 // 32) + ["diag"] * 4
kinds = np.array(kinds[:len(st)])
top = st[:, 1] - st[:, 0]
body = st[:, 2] - st[:, 1]
gap = st[1:, 0] - st[:-1, 2]
print("stages stamped", len(st), "total cycles", st[-1, 2] - st[0, 0])
for k in ("reg", "diag"):
    sel = kinds == k
    print(f"{k:5s} n={sel.sum():5d} top(wait+barrier) mean={top[sel].mean():8.1f} p90={np.percentile(top[sel],90):8.1f}  body mean={body[sel].mean():8.1f}")
print("between-stage gap mean", gap.mean(), " (cursor advance etc.)")
# block transitions: gap after the 4th diag stage
idx = np.where((kinds[:-1] == "diag") & (kinds[1:] == "reg"))[0]
print("diag->reg transition gap mean", gap[idx].mean() if len(idx) else None)
print("sum top", top.sum(), "sum body", body.sum(), "sum gaps", gap.sum())
reg = kinds == "reg"
r = st[reg]
print("regular stage split (cycles, mean): lgkm-wait", (r[:, 3] - r[:, 0]).mean(), " vmcnt+barrier", (r[:, 1] - r[:, 3]).mean(),
      " reads0+deferred MFMAs", (r[:, 4] - r[:, 1]).mean(), " k-steps 0..6", (r[:, 2] - r[:, 4]).mean())
ridx = np.where(reg[:-1] & reg[1:])[0]
print("reg->reg loop overhead (end of body -> next stamp 0)", (st[ridx + 1, 0] - st[ridx, 2]).mean())
