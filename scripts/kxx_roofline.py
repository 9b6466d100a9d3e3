"""K(X,X) assembly rate at the north-star size (16384 points, d=3): device time from the library's hipEvents."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
import ctypes
ctx = _lib.Context.get(0)
for n in (4096, 16384):
    rng = np.random.default_rng(n)
    X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
    y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
    m = HipGaussianProcess(X, y, context=ctx)
    ctx.set_profiling(True); ctx.reset_timers()
    for _ in range(5):
        _lib.check(_lib.load().cbo_gp_fit(m._handle, None, None))
    t = ctx.timers(); ctx.set_profiling(False)
    nt = n // 64
    b = nt * (nt + 1) // 2 * 64 * 64 * 8
    ms = t["ms_kxx"] / 5
    print(f"n={n}: kxx {ms:.3f} ms  {b/ms/1e6:.0f} GB/s ({b/ms/1e6/8000:.2%} of 8 TB/s; upper-tile bytes {b/1e6:.0f} MB)  chol {t['ms_chol']/5:.2f} ms")
    m.close()
