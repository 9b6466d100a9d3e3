"""What the parity tests actually enforce, measured per golden fixture on the GPU: relative error of the
HIP path and of the fp64 oracle against the 80-bit arbiter, the bound the tests apply (rtol + slack x oracle error),
and the HIP-vs-oracle difference, for posterior mean (shifted by 3 max|y| as in the test), variance and acquisition.
usage: python scripts/tolerance_report.py [--large]   (--large adds the 16384-observation configs c4 / c5 on bench.py's data)"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import FIXTURES, load_fixture  # noqa: E402
from cbo_with_oop_amd import CausalExpectedImprovement  # noqa: E402
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float).reshape(-1), np.asarray(b, float).reshape(-1)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


print(f"{'fixture':16s} {'n':>5s} {'m':>6s} {'cond-ish':>9s} | {'mean: hip/truth':>15s} {'oracle/truth':>12s} {'bound':>9s} |"
      f" {'var: hip/truth':>14s} {'oracle/truth':>12s} {'bound':>9s} | {'acq: hip/truth':>14s} {'oracle/truth':>12s} {'bound':>9s} | argmax")
for name in FIXTURES:
    f = load_fixture(name)
    ls = f["lengthscale_arg"]
    kw = dict(variance=float(f["variance"]), lengthscale=ls, ard=not np.isscalar(ls), noise_var=float(f["noise_var"]))
    if f["mX"] is not None:
        lut_m = {**{tuple(r): v for r, v in zip(map(tuple, f["X"]), f["mX"][:, 0])},
                 **{tuple(r): v for r, v in zip(map(tuple, f["Xs"]), f["mXs"][:, 0])}}
        lut_v = {**{tuple(r): v for r, v in zip(map(tuple, f["X"]), f["vX"][:, 0])},
                 **{tuple(r): v for r, v in zip(map(tuple, f["Xs"]), f["vXs"][:, 0])}}
        kw["mean_function"] = lambda a: np.array([[lut_m[tuple(r)]] for r in a])
        kw["variance_adjustment"] = lambda a: np.array([[lut_v[tuple(r)]] for r in a])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        m = HipGaussianProcess(f["X"], f["y"], **kw)
    res = CausalExpectedImprovement(float(f["y_best"]), f["task"], m).sweep(f["Xs"], cost=float(f["cost"]), want_acq=True,
                                                                             want_posterior=True)
    scale = 3 * np.max(np.abs(f["y"]))
    L, _ = m.posterior_state()
    d = np.diag(L)
    out = [f"{name:16s} {f['X'].shape[0]:5d} {f['Xs'].shape[0]:6d} {(d.max() / d.min()) ** 2:9.1e} |"]
    ei_truth = O.expected_improvement(f["mean_truth"], f["var_truth"], float(f["y_best"]), f["task"]) / float(f["cost"])
    big = np.abs(f["acq"][:, 0]) > 1e-6 * np.max(np.abs(f["acq"]))
    for hip, orc, tru in ((res["mean"] + scale, f["mean"] + scale, f["mean_truth"] + scale), (res["var"], f["var"], f["var_truth"]),
                          (res["acq"][big], f["acq"][big], ei_truth[big])):
        o = rel(orc, tru)
        out.append(f" {rel(hip, tru):15.2e} {o:12.2e} {1e-5 + 8 * o:9.2e} |")
    out.append(f" {'same' if res['best_idx'] == int(f['best_idx']) else 'DIFFERS'}")
    print("".join(out))
    m.close()

if "--large" in sys.argv:
    # The two 16384-observation configurations on bench.py's own data (every 128th candidate of the rank's shard plus the device's 64 best).  The
    # 80-bit restatement is not affordable there; the arbiter of the mean is oracle/truth.py:refined_mean (exact kernel
    # entries, iterative refinement with long-double residuals), the acquisition is taken from the arbiter's mean and the
    # oracle's variance, and the variance is quoted against the oracle itself.
    import bench
    from oracle.truth import refined_mean
    print()
    print(f"{'config (device path)':26s} {'tries':>5s} | {'mean / max|y|: dev-arb':>22s} {'oracle-arb':>10s} {'dev-oracle':>10s} |"
          f" {'var: dev-oracle abs':>19s} {'rel':>9s} | {'acq / max|acq|: dev-arb':>23s} {'oracle-arb':>10s} | argmax")
    for cname in ("c4", "c5"):
        X, y, Xs, grid, note = bench.make_problem(bench.CONFIGS[cname], 1, "strong", False)
        y_best, cost = float(y.min()), 3.0
        post = O.fit(X, y)
        # every 128th candidate of the shard AND the fp64 device path's 64 best: on these boxes the acquisition is ~1e-123 of
        # its maximum almost everywhere (the winner sits in a corner of the grid), and a subsample without the top would
        # compare zeros in the acquisition column
        m0 = HipGaussianProcess(X, y)
        a0 = CausalExpectedImprovement(y_best, "min", m0).sweep(Xs, cost=cost, want_acq=True)["acq"][:, 0]
        m0.close()
        top64 = np.argsort(-a0, kind="stable")[:64]
        sub = np.unique(np.concatenate([np.arange(0, Xs.shape[0], 128), top64]))
        acq, _, _, mu, var = O.acquisition_sweep(post, Xs[sub], y_best, cost=cost)
        tm, _ = refined_mean(post, Xs[sub], exact_entries=True)
        acq_t = O.expected_improvement(tm, var, y_best, "min", 0.0) / cost
        scale = np.max(np.abs(y))
        for dtype in (("f64",) if cname == "c4" else ("f64", "f32")):
            m = HipGaussianProcess(X, y, dtype=dtype)
            res = CausalExpectedImprovement(y_best, "min", m).sweep(Xs, cost=cost, want_acq=True, want_posterior=True)
            amax = np.max(np.abs(res["acq"]))
            o_best = int(sub[np.argmax(acq[:, 0])])
            d_best = int(sub[np.argmax(res["acq"][sub, 0])])
            print(f"{cname + ' bench data (' + dtype + ')':26s} {m.jitter_tries:5d} | {np.max(np.abs(res['mean'][sub] - tm)) / scale:22.2e} "
                  f"{np.max(np.abs(mu - tm)) / scale:10.2e} {np.max(np.abs(res['mean'][sub] - mu)) / scale:10.2e} | "
                  f"{np.max(np.abs(res['var'][sub] - var)):19.2e} {np.max(np.abs(res['var'][sub] - var) / var):9.2e} | "
                  f"{np.max(np.abs(res['acq'][sub] - acq_t)) / amax:23.2e} {np.max(np.abs(acq - acq_t)) / amax:10.2e} | "
                  f"{'same' if o_best == d_best else 'DIFFERS'} (subsample + the device's top 64: {int(np.sum(np.abs(acq[:, 0]) > 1e-6 * np.abs(acq).max()))} "
                  f"candidates with an acquisition above 1e-6 of the maximum)")
            m.close()
