"""Timing-only (diagnostic build): per-stage s_memtime stamps of workgroup 0 / wave 0 of the sweep's strip TRSM.
Run with CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so.  Prints, for regular and diagonal stages, the mean cycles spent
(a) waiting at the top (s_waitcnt + barrier) and (b) in the stage body."""
import ctypes, os, sys
os.environ.setdefault("CBO_HIP_STRIP_MASK", "256")        # the two-waves-per-SIMD kernel stamps only when asked
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
kinds, dm = [], []
for i0 in range(0, n, 128):
    kinds += ["reg"] * (i0 // 32) + ["diag"] * 4
    dm += [-1] * (i0 // 32) + [0, 1, 2, 3]
kinds, dm = np.array(kinds[:len(st)]), np.array(dm[:len(st)])
if os.environ.get("CBO_HIP_STRIP_FORM", "8") != "4":
    # two waves per SIMD (trsm_strip8_kernel): wave 0 (upper row half) stamps slots 0..2, wave 4 (lower half) slots 4..6
    print("stages stamped", len(st), "total cycles (wave 0)", st[-1, 2] - st[0, 0])
    for name, o in (("upper half (wave 0)", 0), ("lower half (wave 4)", 4)):
        top, body = st[:, o + 1] - st[:, o], st[:, o + 2] - st[:, o + 1]
        gap = st[1:, o] - st[:-1, o + 2]
        sel = kinds == "reg"
        print(f"{name}: regular stages n={sel.sum()} top(wait+barrier) mean {top[sel].mean():.0f} p90 {np.percentile(top[sel], 90):.0f}"
              f"  body mean {body[sel].mean():.0f} p90 {np.percentile(body[sel], 90):.0f}  gap to next stage mean {gap[sel[:-1]].mean():.0f}")
        for m in range(4):
            sel = dm == m
            issue = st[:, o + 3] - st[:, o + 1]
            print(f"    diagonal stage {m}: top {top[sel].mean():.0f}  body {body[sel].mean():.0f}"
                  f"  (of it DMA issue + ahead loads + cursor: {issue[sel].mean():.0f})")
    per = (st[1:, 0] - st[:-1, 0])
    print("stage period (wave 0 top to next top): regular mean", per[(kinds == "reg")[:-1]].mean(),
          " diagonal by m", [float(per[(dm == m)[:-1]].mean()) for m in range(4)])
    print("sum of periods: regular", per[(kinds == "reg")[:-1]].sum(), " diagonal", per[(kinds == "diag")[:-1]].sum())
    # regular stages by position in their block: the first ones after a diagonal phase, the last ones before the next
    jpos, jend = [], []
    for i0 in range(0, n, 128):
        nst_b = i0 // 32
        jpos += list(range(nst_b)) + [-1] * 4
        jend += [nst_b - 1 - j for j in range(nst_b)] + [-1] * 4
    jpos, jend = np.array(jpos[:len(st)])[:-1], np.array(jend[:len(st)])[:-1]
    big = np.array([i0 >= 1024 for i0 in range(0, n, 128) for _ in range(i0 // 32 + 4)][:len(st)])[:-1]
    print("regular stage period by position (blocks from row 1024 on): first six",
          [int(per[big & (jpos == j)].mean()) for j in range(6)], " last six",
          [int(per[big & (jend == j)].mean()) for j in range(5, -1, -1)], " middle", int(per[big & (jpos >= 6) & (jend >= 6)].mean()))
    sys.exit(0)
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
