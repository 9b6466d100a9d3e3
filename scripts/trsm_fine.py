"""Timing-only (diagnostic build, CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so): s_memtime stamps inside the
four diagonal stages of the block at row 2048, solver wave of column group 0, workgroup 0 of trsm_strip8_kernel.
Points: 0 stage entered, 1 LDS operands in registers, 2 first four MFMAs issued, 3 DMA issued, 4 x ready, 5 x published
(+ rendezvous), 6 y ready, 7 y published (+ rendezvous), 8 stores and remaining updates issued."""
import ctypes, os, sys
os.environ["CBO_HIP_STRIP_MASK"] = "512"
os.environ["CBO_HIP_SWEEP_CACHE"] = "0"
os.environ.setdefault("CBO_HIP_SWEEP", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib, CausalExpectedImprovement
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
n, M = 4096, 16384
rng = np.random.default_rng(0)
X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
Xs = rng.uniform([-5, -5, -5], [5, 20, 5], (M, 3))
m = HipGaussianProcess(X, y)
ei = CausalExpectedImprovement(float(y.min()), "min", m)
for _ in range(3):
    ei.sweep(Xs, cost=3.0)
lib = ctypes.CDLL(os.environ["CBO_HIP_LIB"])
buf = (ctypes.c_ulonglong * 64)()
lib.cbo_diag_trsm_fine(buf)
st = np.frombuffer(buf, dtype=np.uint64).reshape(4, 16).astype(np.int64)
for mm in range(4):
    d = st[mm, :9] - st[mm, 0]
    print(f"diagonal stage {mm}: cycles from stage entry at points 0..8:", d.tolist())
