"""alpha = L^-T z (GPy's woodbury_vector) after a fit, timed from the library's phase timer, for the one-launch chain of
workgroups and for the per-block launches (CBO_HIP_VEC_SOLVE_FORM=1); plus the append-only trial step, whose forward
solve is the mirror image.  usage: python scripts/vec_solve_timing.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]
ctx = _lib.Context.get(0)
rng = np.random.default_rng(0)
for n in sizes:
    X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
    y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
    m = HipGaussianProcess(X, y, noise_var=1e-2)
    m.posterior_state()
    ts = []
    for _ in range(5):
        m.set_data(X, y)                       # refit: alpha is materialised again on the next request
        ctx.set_profiling(True)
        ctx.reset_timers()
        t0 = time.perf_counter()
        L, alpha = m.posterior_state()
        ts.append(ctx.timers()["ms_alpha"])
        ctx.set_profiling(False)
    print(f"n={n}: alpha = L^-T z {np.median(ts):.3f} ms (form {os.environ.get('CBO_HIP_VEC_SOLVE_FORM', 'chain')})")
    m.close()
