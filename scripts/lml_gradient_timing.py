"""Hyper-parameter MLE at N=4096, d=3 (f2): device time of one likelihood + analytic-gradient evaluation against the
refits a finite-difference gradient needs, and a whole optimize() call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
rng = np.random.default_rng(0)
for n, ard in ((1024, False), (4096, False), (4096, True), (8192, False)):
    X = rng.uniform(-3, 3, (n, 3)); y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    m = HipGaussianProcess(X, y, lengthscale=1.0, noise_var=1e-2, ard=ard)
    ctx = _lib.Context.get()
    def timed(fn, reps=5):
        fn(); ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        ctx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    t_fit = timed(m._fit)
    t_grad = timed(m.log_likelihood_gradients)
    p = 1 + m.lengthscale.size + 1
    t0 = time.perf_counter(); res = m.optimize(); t_opt = time.perf_counter() - t0
    print(f"N={n} ard={ard}: refit {t_fit:.2f} ms, likelihood gradients {t_grad:.2f} ms -> one evaluation {t_fit+t_grad:.2f} ms "
          f"(finite differences: {p+1} refits = {(p+1)*t_fit:.2f} ms); optimize(): {res.nfev} evaluations, {t_opt*1e3:.0f} ms, "
          f"lml {-res.fun:.3f}", flush=True)
    m.close()
