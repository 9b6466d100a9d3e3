"""Cholesky (cbo_gp_fit) timing by hipEvents at a few sizes: phases K(X,X) / factorisation, TFLOP/s of the n^3/3 flops.
usage: python scripts/chol_timing.py [n ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbo_with_oop_amd import _lib  # noqa: E402
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess  # noqa: E402

ctx = _lib.Context.get(0)
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384]:
    rng = np.random.default_rng(0)
    X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
    y = np.sin(X).sum(1, keepdims=True)
    m = HipGaussianProcess(X, y, context=ctx, noise_var=1e-2, fit=False)
    _lib.check(lib.cbo_gp_fit(m._handle, None, None))
    reps = 5 if n <= 8192 else 3
    ctx.set_profiling(True)
    ctx.reset_timers()
    for _ in range(reps):
        _lib.check(lib.cbo_gp_fit(m._handle, None, None))
    t = ctx.timers()
    ctx.set_profiling(False)
    n_pad = -(-n // 128) * 128
    ms = t["ms_chol"] / reps
    print(f"n={n}: chol {ms:.3f} ms = {n_pad ** 3 / 3 / ms / 1e9:.1f} TFLOP/s ({ms / (n_pad / 128) * 1e3:.1f} us per panel); "
          f"kxx {t['ms_kxx'] / reps:.3f} ms", flush=True)
    m.close()
