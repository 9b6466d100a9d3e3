"""Timing-only (diagnostic build, CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so): s_memtime stamps of the last diagonal-block
launch of a factorisation -- per interval, wave 0 (loads + row tile + update | tile factor | stores) and waves 1-3
(own panel tiles | rendezvous wait | trailing tiles).  s_memtime counts shader cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
y = np.sin(X).sum(1, keepdims=True)
m = HipGaussianProcess(X, y, noise_var=1e-2)
lib = _lib.load()
if len(sys.argv) > 2:                    # timing-only knob of the block factorisation (kernels_chol.hip: g_diag_knob)
    lib.cbo_diag_set_knob.argtypes = [ctypes.c_int]
    lib.cbo_diag_set_knob(int(sys.argv[2]))
for _ in range(3):
    _lib.check(lib.cbo_gp_fit(m._handle, None, None))
_lib.Context.get(0).synchronize()
buf = (ctypes.c_ulonglong * (4 * 9 * 4))()
lib.cbo_diag_chol_stamps.argtypes = [ctypes.c_void_p]
lib.cbo_diag_chol_stamps(buf)
st = np.frombuffer(buf, dtype=np.uint64).reshape(4, 9, 4).astype(np.int64)
t0 = st[:, 8, 0].min()
print("shader cycles, relative to the first wave's start")
print("kernel: start", st[:, 8, 0] - t0, "block loaded", st[:, 8, 1] - t0, "end", st[:, 8, 2] - t0)
for jb in range(8):
    w0 = st[0, jb] - t0
    print(f"jb {jb}: wave0 start {w0[0]:5d} mfma {w0[1]-w0[0]:4d} factor {w0[2]-w0[1]:4d} stores {w0[3]-w0[2]:4d} | ", end="")
    for w in (1, 2, 3):
        a = st[w, jb] - t0
        print(f"w{w} start {a[0]:5d} panel {a[1]-a[0]:4d} wait {a[2]-a[1]:4d} trail {a[3]-a[2]:4d} | ", end="")
    print()
