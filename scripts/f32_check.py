"""fp32 sweep against the fp64 device sweep and the fp64 oracle on coral-graph ranges (BASELINE.json configs[4]);
prints the errors the stated tolerances of tests/test_f32_gpu.py come from, and the sweep timings.
usage: python scripts/f32_check.py [n_obs ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbo_with_oop_amd import CandidateGrid, CausalExpectedImprovement, _lib  # noqa: E402
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess  # noqa: E402
from cbo_with_oop_amd.graphs import CoralGraph, meshgrid_candidates  # noqa: E402


def problem(n, seed=0, names=("N", "O", "T")):
    box = CoralGraph.bounds(list(names))
    lo, hi = np.array([b[0] for b in box], float), np.array([b[1] for b in box], float)
    rng = np.random.default_rng(seed)
    X = rng.uniform(lo, hi, (n, len(box)))
    u = (X - lo) / (hi - lo)
    y = (np.sin(3 * u[:, 0]) + np.cos(2 * u[:, 1]) * u[:, 2] + 0.05 * rng.standard_normal(n))[:, None]
    return box, X, y


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [500, 2048, 4096]
    ctx = _lib.Context.get(0)
    print("selftest", ctx.selftest_mfma())
    for n, noise in [(n, nv) for n in sizes for nv in (1e-10, 1e-2)]:
        box, X, y = problem(n)
        Xs = meshgrid_candidates(box, (32, 32, 16))
        y_best = float(y.min())
        m64 = HipGaussianProcess(X, y, context=ctx, noise_var=noise)
        m32 = HipGaussianProcess(X, y, context=ctx, noise_var=noise, dtype="f32")
        r64 = CausalExpectedImprovement(y_best, "min", m64).sweep(Xs, cost=3.0, want_acq=True, want_posterior=True)
        r32 = CausalExpectedImprovement(y_best, "min", m32).sweep(Xs, cost=3.0, want_acq=True, want_posterior=True)
        scale = np.max(np.abs(y))
        dm = np.max(np.abs(r32["mean"] - r64["mean"])) / scale
        dv = np.max(np.abs(r32["var"] - r64["var"]))
        dvr = np.max(np.abs(r32["var"] - r64["var"]) / r64["var"])
        amax = np.max(np.abs(r64["acq"]))
        da = np.max(np.abs(r32["acq"] - r64["acq"])) / amax
        order = np.argsort(-np.abs(r32["acq"][:, 0]) if False else r32["acq"][:, 0])[::-1]
        rank = int(np.where(order == r64["best_idx"])[0][0])
        print(f"n={n} noise={noise:g}: jitter tries {m64.jitter_tries}; var range [{r64['var'].min():.3e}, {r64['var'].max():.3e}]; "
              f"mean err/scale {dm:.3e}; var abs err {dv:.3e} (rel {dvr:.3e}); acq err/max {da:.3e}; "
              f"best64 {r64['best_idx']} best32 {r32['best_idx']} rank of best64 in f32 order {rank}; "
              f"best acq rel diff {abs(r32['best_val'] - r64['best_val']) / max(abs(r64['best_val']), 1e-300):.3e}")
        g64, g32 = CandidateGrid(Xs, m64, context=ctx), CandidateGrid(Xs, m32, context=ctx)
        for name, m, g in (("f64", m64, g64), ("f32", m32, g32)):
            ei = CausalExpectedImprovement(y_best, "min", m)
            os.environ["CBO_HIP_SWEEP_CACHE"] = "0"
            ei.sweep(g, cost=3.0)
            ctx.set_profiling(True)
            ctx.reset_timers()
            for _ in range(3):
                m._fit()
                ei.sweep(g, cost=3.0)
            t = ctx.timers()
            ctx.set_profiling(False)
            fl = t["trsm_flops"] / max(1, t["n_trsm_launches"])
            ms = t["ms_trsm"] / max(1, t["n_trsm_launches"])
            print(f"   {name}: trsm {ms:.3f} ms/launch = {fl / ms / 1e9:.1f} TFLOP/s; kstar {t['ms_kstar'] / 3:.3f} ms; "
                  f"chol {t['ms_chol'] / 3:.3f} ms; convert {t['ms_f32_convert'] / 3:.3f} ms; launches {t['n_trsm_launches']}")
        g64.close(); g32.close(); m64.close(); m32.close()


if __name__ == "__main__":
    main()
