"""What the parity tests actually enforce, measured per golden fixture on the GPU: relative error of the
HIP path and of the fp64 oracle against the 80-bit arbiter, the bound the tests apply (rtol + slack x oracle error),
and the HIP-vs-oracle difference, for posterior mean (shifted by 3 max|y| as in the test), variance and acquisition.
usage: python scripts/tolerance_report.py"""
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
