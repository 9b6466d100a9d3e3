"""sha256 of the factor and of alpha after a fit at the given sizes (same seeds as the full-size form test): two builds or
two schedules that claim the same bits print the same line.  usage: factor_digest.py [n ...]"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]:
    rng = np.random.default_rng(n)
    X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
    y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
    m = HipGaussianProcess(X, y)
    L, alpha = m.posterior_state()
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(L).tobytes())
    h.update(np.ascontiguousarray(alpha).tobytes())
    print(f"n={n} tries={m.jitter_tries} digest {h.hexdigest()[:24]}", flush=True)
    m.close()
