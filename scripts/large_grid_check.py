"""Very large candidate grids (up to a million candidates, refit + sweep overlapped where it applies): shapes, index
arithmetic and a subsample against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import CausalExpectedImprovement, CandidateGrid
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from oracle import gp_oracle as O
rng = np.random.default_rng(1)
for n, m in ((300, 1_000_000), (4096, 262_144), (1500, 500_000)):
    X = rng.uniform(-3, 3, (n, 3)); y = np.sin(X.sum(1, keepdims=True)) + 0.05 * rng.standard_normal((n, 1))
    Xs = rng.uniform(-3, 3, (m, 3))
    model = HipGaussianProcess(X, y, noise_var=1e-2, fit=False)
    g = CandidateGrid(Xs, model)
    t0 = time.perf_counter()
    r = CausalExpectedImprovement(float(y.min()), "min", model).sweep(g, want_acq=True, want_posterior=True)
    dt = time.perf_counter() - t0
    sub = rng.choice(m, 300, replace=False)
    sub = np.unique(np.concatenate([sub, [r["best_idx"], m - 1, 0]]))
    post = O.fit(X, y, noise_var=1e-2)
    mu, var = O.predict(post, Xs[sub])
    ok = np.allclose(r["mean"][sub], mu, rtol=1e-6, atol=1e-8) and np.allclose(r["var"][sub], var, rtol=1e-6, atol=1e-10)
    print(f"N={n} M={m}: {dt*1e3:.1f} ms, argmax consistent {int(np.argmax(r['acq'][:,0])) == r['best_idx']}, subsample vs oracle {ok}", flush=True)
    g.close(); model.close()
