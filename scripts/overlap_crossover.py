"""Where does overlapping refit and sweep pay?  Step time of cbo_gp_fit_sweep against the two calls over a grid
of (observations, candidates), d=3."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
lib, ctx = _lib.load(), _lib.Context.get()
rng = np.random.default_rng(0)
bv, bi = ctypes.c_double(), ctypes.c_int64()
if len(sys.argv) > 1:
    sizes = [tuple(int(t) for t in a.split('x')) for a in sys.argv[1:]]
else:
  sizes = [(512, 16384), (1024, 16384), (2048, 16384), (4096, 200), (4096, 4096), (4096, 12288), (4096, 16384), (4096, 20480), (4096, 32768), (4096, 65536), (8192, 16384), (8192, 65536)]
for n, mm in sizes:
    X = rng.uniform(-5, 5, (n, 3)); y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    Xs = rng.uniform(-5, 5, (mm, 3))
    m = HipGaussianProcess(X, y, noise_var=1e-2, fit=False)
    g = CandidateGrid(Xs, m)
    def fused():
        _lib.check(lib.cbo_gp_fit_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None, ctypes.byref(bv), ctypes.byref(bi), None, None))
    def two():
        _lib.check(lib.cbo_gp_fit(m._handle, None, None))
        _lib.check(lib.cbo_acq_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None, ctypes.byref(bv), ctypes.byref(bi)))
    def sweep_only():
        _lib.check(lib.cbo_acq_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None, ctypes.byref(bv), ctypes.byref(bi)))
    out = []
    for fn in (fused, two, sweep_only):
        fn(); fn(); ctx.synchronize()
        reps = 5 if n <= 4096 else 2
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        ctx.synchronize()
        out.append((time.perf_counter() - t0) / reps * 1e3)
    print(f"N={n:6d} M={mm:6d}: fit_sweep {out[0]:8.2f} ms, two calls {out[1]:8.2f} ms, ratio {out[1]/out[0]:.3f}; sweep alone {out[2]:8.2f} ms", flush=True)
    g.close(); m.close()
