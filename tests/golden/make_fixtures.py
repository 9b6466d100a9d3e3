"""Generates the golden fixtures under tests/golden/ (run once in the build container; commit the
.npz files).  Inputs are data only:

* ``interventional_data_{x,y}_BO.npy`` of toy_graph and complete_graph from /root/reference/data --
  plain float64 arrays, loaded with ``numpy.load(allow_pickle=False)``.  Every other .npy/.pkl in the
  reference's data/ is a pickled object array / DataFrame: the safe loader refuses them and they are
  NOT used (no unpickling of reference files).
* the toy initial interventional sets recorded in SURVEY.md §8c item 1 (values captured from the
  reference's own ``define_initial_data_cbo`` in the survey session), rebuilt here from the closed
  form of the toy SEM (ToyGraph, verified to 4e-14 on the BO file) and the recorded permutation.

Expected outputs come from the numpy/scipy oracle (oracle/gp_oracle.py, "parity unpinned": the
reference holds no output vectors for this path) and from the 80-bit arbiter (oracle/gp_truth_ld.c).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cbo_with_oop_amd.graphs import CompleteGraph, CoralGraph, ToyGraph, meshgrid_candidates  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from oracle.truth import truth_predict  # noqa: E402

REF = "/root/reference/data"


def case(name, X, y, Xs, y_best, task="min", cost=1.0, mX=None, vX=None, mXs=None, vXs=None, noise_var=1e-10,
         variance=1.0, lengthscale=1.0, note=""):
    post = O.fit(X, y, mX, vX, variance, lengthscale, noise_var)
    acq, best_val, best_idx, mu, var = O.acquisition_sweep(post, Xs, y_best, mXs, vXs, task, cost)
    _, var_w = O.predict(post, Xs, mXs, vXs, var_form="woodbury")
    mt, vt, at = truth_predict(X, y, Xs, mX, vX, mXs, vXs, variance, lengthscale,
                               diag_add=noise_var + O.GPY_DIAG_JITTER + post.jitter, noise_var=noise_var)
    out = dict(X=X, y=y, Xs=Xs, y_best=np.float64(y_best), task=np.array(task), cost=np.float64(cost),
               noise_var=np.float64(noise_var), variance=np.float64(variance),
               lengthscale=np.atleast_1d(np.asarray(lengthscale, dtype=np.float64)),
               L=post.L, alpha=post.alpha, jitter=np.float64(post.jitter), tries=np.int64(post.tries),
               mean=mu, var=var, var_woodbury=var_w, acq=acq, best_val=np.float64(best_val),
               best_idx=np.int64(best_idx), mean_truth=mt, var_truth=vt, alpha_truth=at, note=np.array(note))
    for k, v in dict(mX=mX, vX=vX, mXs=mXs, vXs=vXs).items():
        if v is not None:
            out[k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    srt = np.sort(acq[:, 0])[::-1]
    print(f"{name}: N={X.shape[0]} d={X.shape[1]} M={Xs.shape[0]} tries={post.tries} best_idx={best_idx} "
          f"best={best_val:.6e} gap={(srt[0]-srt[1])/abs(srt[0]):.2e} "
          f"var_err_oracle={np.max(np.abs(var-vt)/vt):.2e} var_err_woodbury={np.max(np.abs(var_w-vt)/vt):.2e}")


def main():
    rng = np.random.default_rng(20250204)

    # 1. real reference data: toy joint (X,Z) BO grid, d=2
    X = np.load(f"{REF}/toy_graph/interventional_data_x_BO.npy", allow_pickle=False)
    y = np.load(f"{REF}/toy_graph/interventional_data_y_BO.npy", allow_pickle=False)
    assert np.max(np.abs(y[:, 0] - ToyGraph.target_do_z(X[:, 1]))) < 1e-12       # toy SEM identity
    Xs = meshgrid_candidates(ToyGraph.bounds(["X", "Z"]), [16, 16])
    case("toy_bo_d2", X, y, Xs, y.min(), cost=2.0, note="reference data/toy_graph/*_BO.npy, non-causal GP")

    # 2. real reference data: complete graph joint (B,E,D) BO grid, d=3
    X = np.load(f"{REF}/complete_graph/interventional_data_x_BO.npy", allow_pickle=False)
    y = np.load(f"{REF}/complete_graph/interventional_data_y_BO.npy", allow_pickle=False)
    # the 20 data points lie on a line; a box grid is mostly prior plateau (EI equal to 1e-14), so the
    # candidates hug the line instead: 21 stations x 24 seeded offsets of length <= 1.2
    lo_, hi_ = X.min(0), X.max(0)
    stations = lo_ + np.linspace(0, 1, 21)[:, None] * (hi_ - lo_)
    offs = rng.standard_normal((24, 3))
    offs *= (rng.uniform(0.05, 1.2, (24, 1)) / np.linalg.norm(offs, axis=1, keepdims=True))
    Xs = np.ascontiguousarray((stations[:, None, :] + offs[None, :, :]).reshape(-1, 3))
    case("complete_bo_d3", X, y, Xs, y.min(), cost=3.0, note="reference data/complete_graph/*_BO.npy")

    # 3. toy initial interventional sets: 10 points per set, 200-candidate sweep (C1 shape)
    perm = [18, 1, 19, 8, 10, 17, 6, 13, 4, 2]
    gx, gz = np.linspace(-4, 3, 20), np.linspace(-4, 18, 20)
    x0 = gx[perm][:, None]
    z0 = gz[perm][:, None]
    assert np.allclose(x0[:3, 0], [2.63158, -3.63158, 3.0], atol=5e-6)
    assert np.allclose(z0[:3, 0], [16.84211, -2.84211, 18.0], atol=5e-6)
    yx, yz = ToyGraph.target_do_x(x0), ToyGraph.target_do_z(z0)
    opt_y = min(yx.min(), yz.min())
    assert abs(opt_y - (-2.1081858288678093)) < 1e-9, opt_y                      # value captured from the reference
    case("toy_init_X", x0, yx, meshgrid_candidates(ToyGraph.bounds(["X"]), [200]), opt_y,
         note="toy initial set ['X'] (SURVEY §8c.1), C1-shaped 200-candidate sweep")
    case("toy_init_Z", z0, yz, meshgrid_candidates(ToyGraph.bounds(["Z"]), [200]), opt_y,
         note="toy initial set ['Z'] (SURVEY §8c.1)")

    # 4. C1: toy, 50 observation points on the Z set (ill-conditioned: 50 points, lengthscale 1)
    z = np.concatenate([z0[:, 0], rng.uniform(-5, 20, 40)])[:, None]
    yz = ToyGraph.target_do_z(z)
    case("toy_c1_Z50", z, yz, meshgrid_candidates(ToyGraph.bounds(["Z"]), [200]), yz.min(),
         note="BASELINE config 1 shape: 50 obs, 200 candidates, d=1")

    # 5. causal GP: synthetic prior mean / variance vectors (what DoCalculus closures return)
    N, M = 40, 300
    X = rng.uniform([-5, -5], [4, 5], (N, 2))
    f = lambda a: np.sin(a[:, :1]) + 0.3 * a[:, 1:] ** 2 / 5.0
    pm = lambda a: 0.8 * f(a) + 0.1
    pv = lambda a: 0.05 + 0.02 * np.cos(a[:, :1]) ** 2
    y = f(X) + 0.05 * rng.standard_normal((N, 1))
    Xs = rng.uniform([-5, -5], [4, 5], (M, 2))
    case("causal_d2", X, y, Xs, y.min(), cost=2.0, mX=pm(X), vX=pv(X), mXs=pm(Xs), vXs=pv(Xs),
         note="CausalRBF + mean function with synthetic m(.), v(.) vectors")

    # 6. task max (reference returns -EI with the same u), coral-like ranges, d=3 well conditioned
    b = CoralGraph.bounds(["N", "O", "C"])
    lo, hi = np.array([v[0] for v in b], float), np.array([v[1] for v in b], float)
    X = rng.uniform(lo, hi, (200, 3))
    y = (np.sin(X).sum(1) + 0.1 * rng.standard_normal(200))[:, None]
    case("coral_max_d3", X, y, meshgrid_candidates(b, [10, 10, 10]), y.max(), task="max", cost=3.0,
         note="task='max' quirk; N=200 not a multiple of 64")

    # 7. graph-level GP shape: ARD lengthscales, noise 1e-2, d=4 (create_graph_gp)
    X = rng.uniform(-2, 2, (150, 4))
    y = (np.cos(X[:, 0]) * X[:, 1] + 0.2 * X[:, 2] - 0.1 * X[:, 3] ** 2)[:, None]
    Xs = rng.uniform(-2, 2, (130, 4))
    case("graph_ard_d4", X, y, Xs, y.min(), noise_var=1e-2, variance=1.7, lengthscale=[0.7, 1.3, 2.0, 0.9],
         note="GRAPH_GP: ARD RBF, noise fixed 1e-2")

    # 8. jitter ladder: Ky not positive definite until jitter is added (negative 'noise' cancels the 1e-8)
    X = rng.uniform(-1, 1, (24, 1))
    X = np.vstack([X, X[:8]])                      # exact duplicates -> singular K
    y = np.sin(3 * X) + 0.01 * rng.standard_normal(X.shape)
    case("jitter_ladder", X, y, np.linspace(-1, 1, 64)[:, None], y.min(), noise_var=-1e-8 - 1e-9,
         note="duplicate rows + negative effective diagonal: dpotrf fails, jitchol adds mean(diag)*1e-6")


if __name__ == "__main__":
    main()
