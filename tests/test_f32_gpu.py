"""GPU tests of the fp32 sweep (BASELINE.json configs[4]: coral_graph, "fp32 path with MFMA"), through the C-ABI
(cbo_gp_create(dtype = CBO_DTYPE_F32)), against the fp64 numpy/scipy oracle on the coral graph's interventional
ranges (/root/reference/src/graphs/impl/CoralGraph.py:177-184).

STATED TOLERANCE (fp32 cannot meet the north_star's 1e-5 rtol on the variance; this is what it meets,
measured with scripts/f32_check.py on an MI355X, with a margin of about 4x):

  * posterior mean      the mean of an fp32 model is an fp64 quantity, K*^T alpha (GPy's own formula, which is also the
                        oracle's); two fp64 evaluations of it differ by eps * cond(Ky) * |k*| |alpha|, which on the coral
                        ranges with the reference's 1e-10 noise is anything from 1e-9 to 0.4 (cond(Ky) up to 1e11 on the
                        2-D (N, T) set).  So the mean is held to the arbiter rule of tests/conftest.py:
                        |mean32 - truth| <= 1e-5 * max|y| + 8 * max|oracle - truth|, truth = the 80-bit restatement
                        (oracle/gp_truth_ld.c) on a 257-candidate subsample.
  * posterior variance  |var32 - var|   <= 2e-4 * k(x,x)      absolute, k(x,x) = 1 (measured <= 4.4e-5 with the
                        reference's 1e-10 noise at 4096 points, <= 2e-6 with the graph-level GPs' 1e-2 noise: an fp32
                        triangular solve is accurate to about 6e-8 * sqrt(cond(Ky)) relative in L^-1 k*, and the variance
                        is the cancellation k(x,x) - |L^-1 k*|^2).  Variances below ~1e-4 are therefore NOT resolved: with
                        the reference's 1e-10 noise on its dense 1-D / 2-D sets (true variances 1e-9, EI underflowing to
                        1e-283) the fp32 path is the wrong tool, and those sets are tested at the graph-level GPs' 1e-2.
  * acquisition         |acq32 - acq|   <= 5e-5 * max|acq|    (measured <= 7e-6 up to 4096 observations; at the config-5
                        shard shape, 16384 observations, 2e-4 * max|acq|: measured 8.6e-5 against the fp64 device path)
  * arg-max             the oracle's arg-max is among the fp32 path's top 8 candidates and the fp32 winner's acquisition
                        value is within 1e-4 relative of the oracle's best value (measured: identical arg-max in every case)
"""
import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

MEAN_RTOL, MEAN_SLACK, VAR_TOL, ACQ_TOL, BEST_RTOL, TOP_K = 1e-5, 8.0, 2e-4, 5e-5, 1e-4, 8
ACQ_TOL_FULL = 2e-4       # at the config-5 shard shape (16384 observations): measured 8.6e-5 of max|acq| against the fp64 path


@pytest.fixture(scope="module")
def hip():
    import cbo_with_oop_amd as pkg
    from cbo_with_oop_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: -m gpu tests need an MI355X"
    return pkg


def coral_problem(n, names=("N", "O", "T"), seed=0):
    from cbo_with_oop_amd.graphs import CoralGraph
    box = CoralGraph.bounds(list(names))
    lo, hi = np.array([b[0] for b in box], float), np.array([b[1] for b in box], float)
    rng = np.random.default_rng(seed)
    X = rng.uniform(lo, hi, (n, len(box)))
    u = (X - lo) / (hi - lo)
    y = (np.sin(3 * u[:, 0]) + np.cos(2 * u[:, -1]) * u[:, len(box) // 2] + 0.05 * rng.standard_normal(n))[:, None]
    return box, X, y


def mean_ok(mean32, mu, sub, truth_mean, y):
    """Arbiter rule for the mean (module docstring): returns (passes, message)."""
    scale = np.max(np.abs(y))
    oracle_err = np.max(np.abs(mu[sub] - truth_mean))
    err = np.max(np.abs(mean32[sub] - truth_mean))
    bound = MEAN_RTOL * scale + MEAN_SLACK * oracle_err
    return err <= bound, f"mean: |f32 path - truth| {err:.3e} > {bound:.3e} (oracle - truth {oracle_err:.3e})"


def truth_subsample(X, y, Xs, best_idx, noise, **prior):
    from oracle.truth import truth_predict
    sub = np.unique(np.concatenate([np.linspace(0, Xs.shape[0] - 1, 256).astype(int), [best_idx]]))
    kw = {k: (v if k in ("mX", "vX") else v[sub]) for k, v in prior.items()}
    tm, tv, _ = truth_predict(X, y, Xs[sub], diag_add=noise + 1e-8, noise_var=noise, **kw)
    return sub, tm, tv


def check_against_oracle(res, acq, best_val, best_idx, mu, var, y, sub, truth_mean):
    problems = []
    ok, msg = mean_ok(res["mean"], mu, sub, truth_mean, y)
    if not ok:
        problems.append(msg)
    dv = np.max(np.abs(res["var"] - var))
    if dv > VAR_TOL:
        problems.append(f"var: abs err {dv:.3e} > {VAR_TOL:.1e}")
    amax = np.max(np.abs(acq))
    da = np.max(np.abs(res["acq"] - acq))
    if da > ACQ_TOL * amax:
        problems.append(f"acq: err {da:.3e} > {ACQ_TOL:.1e} * {amax:.3e}")
    top = np.argsort(-res["acq"][:, 0], kind="stable")[:TOP_K]
    if best_idx not in top:
        problems.append(f"oracle arg-max {best_idx} not in the fp32 top {TOP_K}: {top}")
    if abs(res["best_val"] - best_val) > BEST_RTOL * abs(best_val):
        problems.append(f"best value {res['best_val']:.9e} vs {best_val:.9e}")
    if int(np.argmax(res["acq"][:, 0])) != res["best_idx"]:
        problems.append("device arg-max differs from numpy's on the device's own output")
    assert not problems, "; ".join(problems)


@pytest.mark.parametrize("n,names,grid,noise", [
    (1300, ("N", "O", "T"), (16, 16, 16), 1e-10),      # n_pad 1408 -> 1536 rows in the fp32 layout (identity padding)
    (2048, ("N", "O", "T"), (32, 16, 16), 1e-2),       # the graph-level GPs' noise (src/utils_functions/utils.py:40-44)
    (700, ("N", "T"), (64, 64), 1e-2),                 # 1- and 2-D sets: dense data, cond(Ky) ~ 1e11 at the reference's
    (200, ("T",), (200,), 1e-2),                       # 1e-10 noise, predictive variances far below what fp32 resolves
])
def test_f32_sweep_on_coral_ranges_against_the_oracle(hip, n, names, grid, noise):
    from cbo_with_oop_amd import CausalExpectedImprovement
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    box, X, y = coral_problem(n, names)
    Xs = meshgrid_candidates(box, grid)
    m = HipGaussianProcess(X, y, noise_var=noise, dtype="f32")
    y_best, cost = float(y.min()), float(len(names))
    res = CausalExpectedImprovement(y_best, "min", m).sweep(Xs, cost=cost, want_acq=True, want_posterior=True)
    post = O.fit(X, y, noise_var=noise)
    assert m.jitter_tries == post.tries
    acq, best_val, best_idx, mu, var = O.acquisition_sweep(post, Xs, y_best, cost=cost)
    sub, tm, _ = truth_subsample(X, y, Xs, best_idx, noise)
    check_against_oracle(res, acq, best_val, best_idx, mu, var, y, sub, tm)
    # predict() of an fp32 model runs the same path: same bits as the sweep's posterior
    mean, v = m.predict(Xs[:333])
    assert np.array_equal(mean, res["mean"][:333]) and np.array_equal(v, res["var"][:333])


def test_f32_causal_model_task_max_and_chunked_workspace(hip, monkeypatch):
    """Causal kernel (prior mean / variance closures) under task 'max' with the reference's -EI quirk, once with the
    candidates cut into many workspace chunks: chunking must not change a bit."""
    from cbo_with_oop_amd import _lib, CausalExpectedImprovement
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    box, X, y = coral_problem(600, ("O", "T"))
    Xs = meshgrid_candidates(box, (40, 50))
    lo, hi = np.array([b[0] for b in box], float), np.array([b[1] for b in box], float)
    mean_fn = lambda a: 0.3 * np.sin((np.asarray(a)[:, :1] - lo[0]) / (hi[0] - lo[0]) * 2.0)
    var_fn = lambda a: 0.05 + 0.1 * ((np.asarray(a)[:, 1:2] - lo[1]) / (hi[1] - lo[1])) ** 2
    kw = dict(mean_function=mean_fn, variance_adjustment=var_fn, noise_var=1e-2, dtype="f32")
    m = HipGaussianProcess(X, y, **kw)
    y_best = float(y.max())
    res = CausalExpectedImprovement(y_best, "max", m).sweep(Xs, cost=2.0, want_acq=True, want_posterior=True)
    post = O.fit(X, y, mX=mean_fn(X), vX=var_fn(X), noise_var=1e-2)
    acq, best_val, best_idx, mu, var = O.acquisition_sweep(post, Xs, y_best, task="max", cost=2.0, mXs=mean_fn(Xs),
                                                           vXs=var_fn(Xs))
    sub, tm, _ = truth_subsample(X, y, Xs, best_idx, 1e-2, mX=mean_fn(X), vX=var_fn(X), mXs=mean_fn(Xs), vXs=var_fn(Xs))
    ok, msg = mean_ok(res["mean"], mu, sub, tm, y)
    assert ok, msg
    assert np.max(np.abs(res["var"] - var)) <= VAR_TOL * (1.0 + np.max(var_fn(Xs)))
    assert np.max(np.abs(res["acq"] - acq)) <= ACQ_TOL * np.max(np.abs(acq))
    monkeypatch.setenv("CBO_HIP_WORKSPACE_MB", "1")          # 1 MiB / (768 rows * 4 B) -> 320 columns per chunk
    ctx2 = _lib.Context(0)
    m2 = HipGaussianProcess(X, y, context=ctx2, **kw)
    res2 = CausalExpectedImprovement(y_best, "max", m2).sweep(Xs, cost=2.0, want_acq=True, want_posterior=True)
    assert res2["best_idx"] == res["best_idx"]
    for k in ("acq", "mean", "var"):
        assert np.array_equal(res2[k], res[k]), k
    m2.close()
    ctx2.close()


@pytest.mark.parametrize("n,m,d", [(1, 1, 1), (3, 70, 2), (129, 65, 3), (256, 64, 1), (257, 1000, 3)])
def test_f32_ragged_sizes(hip, n, m, d):
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    rng = np.random.default_rng(n * 1000 + m)
    X = rng.uniform(-2, 2, (n, d)) * 3.0
    y = np.sin(X.sum(1, keepdims=True))
    Xs = rng.uniform(-2, 2, (m, d)) * 3.0
    model = HipGaussianProcess(X, y, noise_var=1e-4, dtype="f32")
    mean, var = model.predict(Xs)
    post = O.fit(X, y, noise_var=1e-4)
    mu, v = O.predict(post, Xs)
    assert np.max(np.abs(mean - mu)) <= 1e-7 * max(1.0, np.max(np.abs(y)))       # noise 1e-4: well conditioned
    assert np.max(np.abs(var - v)) <= VAR_TOL


def test_f32_model_refits_and_keeps_fp64_services(hip):
    """set_data refits (and refreshes the fp32 copies); the fp64 services of the model -- posterior export, likelihood,
    prediction gradients -- are untouched by the dtype; append declines (the caller refits)."""
    from cbo_with_oop_amd import CausalExpectedImprovement
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    box, X, y = coral_problem(400, ("N", "O"))
    Xs = meshgrid_candidates(box, (30, 30))
    m32 = HipGaussianProcess(X[:300], y[:300], noise_var=1e-3, dtype="f32")
    m64 = HipGaussianProcess(X[:300], y[:300], noise_var=1e-3)
    assert not m32.append(X[300], y[300])                      # fp32 models decline the append shortcut
    m32.set_data(X, y)
    m64.set_data(X, y)
    post = O.fit(X, y, noise_var=1e-3)
    y_best = float(y.min())
    res = CausalExpectedImprovement(y_best, "min", m32).sweep(Xs, cost=2.0, want_acq=True, want_posterior=True)
    acq, best_val, best_idx, mu, var = O.acquisition_sweep(post, Xs, y_best, cost=2.0)
    sub, tm, _ = truth_subsample(X, y, Xs, best_idx, 1e-3)
    check_against_oracle(res, acq, best_val, best_idx, mu, var, y, sub, tm)
    L32, a32 = m32.posterior_state()
    L64, a64 = m64.posterior_state()
    assert np.array_equal(L32, L64) and np.array_equal(a32, a64)        # the fit is the same fp64 fit
    assert m32.log_likelihood() == m64.log_likelihood()
    g32, g64 = m32.get_prediction_gradients(Xs[:5]), m64.get_prediction_gradients(Xs[:5])
    assert np.array_equal(g32[0], g64[0]) and np.array_equal(g32[1], g64[1])
    # one overlapped-call request on an fp32 model = fit, then fp32 sweep
    fused = CausalExpectedImprovement(y_best, "min", m32).sweep(Xs, cost=2.0, want_acq=True, refit=True)
    assert fused["best_idx"] == res["best_idx"] and np.array_equal(fused["acq"], res["acq"])


def test_f32_config5_shard_at_full_size(hip):
    """BASELINE config 5 at the shape bench.py --config c5 times per GPU: 16384 observations on the coral (N, O, T) box
    (the bench's own seeded data), the first 32768 candidates of the 64^3 grid -- 64 row blocks of the 256-row permuted fp32
    layout, the jitchol retry that this data needs on every fit, alpha by the blocked backward solve.  The fp64 oracle
    (dpotrf of a 16384^2 matrix on the host) checks a 257-candidate subsample and the winners; the fp64 device path (itself
    held to the oracle at this size by test_c4_size_chunked_workspace) checks every candidate.  The 80-bit arbiter is not
    affordable at this size (1.5e12 long-double operations); its stand-in is the exact-entry system solved by iterative
    refinement (oracle/truth.py:refined_mean), pinned against the 80-bit restatement on small cases."""
    from cbo_with_oop_amd import CausalExpectedImprovement
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    box = [(-2.0, 5.0), (2.0, 4.0), (2450.0, 2500.0)]                     # CoralGraph (N, O, T), bench.py CONFIGS["c5"]
    lo, hi = np.array([b[0] for b in box]), np.array([b[1] for b in box])
    n = 16384
    X = np.random.default_rng(0).uniform(lo, hi, (n, 3))
    u = (X - lo) / (hi - lo)
    y = (np.sin(3 * u[:, 0]) + np.cos(2 * u[:, 2]) * u[:, 1] + 0.05 * np.random.default_rng(1).standard_normal(n))[:, None]
    Xs = meshgrid_candidates(box, (64, 64, 64))[:32768]
    y_best, cost = float(y.min()), 3.0
    m32 = HipGaussianProcess(X, y, dtype="f32")
    res = CausalExpectedImprovement(y_best, "min", m32).sweep(Xs, cost=cost, want_acq=True, want_posterior=True)
    assert int(np.argmax(res["acq"][:, 0])) == res["best_idx"]
    # the whole grid against the fp64 device path
    m64 = HipGaussianProcess(X, y)
    ref = CausalExpectedImprovement(y_best, "min", m64).sweep(Xs, cost=cost, want_acq=True, want_posterior=True)
    assert m32.jitter_tries == m64.jitter_tries
    amax = np.max(np.abs(ref["acq"]))
    top32 = np.argsort(-res["acq"][:, 0], kind="stable")[:TOP_K]
    problems = []
    if np.max(np.abs(res["var"] - ref["var"])) > VAR_TOL:
        problems.append(f"var vs fp64 device path: {np.max(np.abs(res['var'] - ref['var'])):.3e} > {VAR_TOL:.1e}")
    if np.max(np.abs(res["acq"] - ref["acq"])) > ACQ_TOL_FULL * amax:
        problems.append(f"acq vs fp64 device path: {np.max(np.abs(res['acq'] - ref['acq'])):.3e} > {ACQ_TOL_FULL:.1e} * {amax:.3e}")
    if np.max(np.abs(res["mean"] - ref["mean"])) > MEAN_RTOL * np.max(np.abs(y)):
        problems.append(f"mean vs fp64 device path: {np.max(np.abs(res['mean'] - ref['mean'])):.3e}")
    # SURVEY.md 7 asks for arg-max EQUALITY at config 5: asserted.  Should an fp32 rounding ever swap two candidates whose
    # fp64 acquisitions are closer than the path's stated tolerance, the message says by how much, and whether the fp64
    # winner is still among the fp32 top 8 (the stated fallback of the module docstring) -- it is a failure either way.
    if ref["best_idx"] != res["best_idx"]:
        a64 = ref["acq"][:, 0]
        margin = (a64[ref["best_idx"]] - a64[res["best_idx"]]) / abs(a64[ref["best_idx"]])
        problems.append(f"fp32 arg-max {res['best_idx']} != fp64 arg-max {ref['best_idx']}: the fp64 path separates them by "
                        f"{margin:.3e} of the winner's value; fp64 winner {'in' if ref['best_idx'] in top32 else 'NOT in'} the "
                        f"fp32 top {TOP_K} {top32}")
    if abs(res["best_val"] - ref["best_val"]) > BEST_RTOL * abs(ref["best_val"]):
        problems.append(f"best value {res['best_val']:.9e} vs {ref['best_val']:.9e}")
    m64.close()
    # a subsample (and both winners) against the oracle
    post = O.fit(X, y)
    assert m32.jitter_tries == post.tries and post.tries >= 1, (m32.jitter_tries, post.tries)   # the retry IS exercised
    sub = np.unique(np.concatenate([np.arange(0, 32768, 128), top32, [ref["best_idx"]]]))
    acq, _, _, mu, var = O.acquisition_sweep(post, Xs[sub], y_best, cost=cost)
    if np.max(np.abs(res["var"][sub] - var)) > VAR_TOL:
        problems.append(f"var vs oracle: {np.max(np.abs(res['var'][sub] - var)):.3e} > {VAR_TOL:.1e}")
    # The mean and the acquisition by the arbiter rule of tests/conftest.py.  The arbiter: the exact-arithmetic kernel on
    # the same inputs (entries from direct coordinate differences), solved by iterative refinement with long-double
    # residuals (oracle/truth.py:refined_mean, pinned against the all-long-double restatement in tests/test_oracle.py).
    # GPy's |x|^2 + |x'|^2 - 2 x.x' loses 1e-9 of every entry at T ~ 2475, which this ill-conditioned system (jitchol's
    # 1e-6 jitter under 16384 points) turns into 1e-3 of the mean -- in the oracle and in the device path alike.
    from oracle.truth import refined_mean
    tm, _ = refined_mean(post, Xs[sub], exact_entries=True)
    oracle_err = np.max(np.abs(mu - tm))
    dev_err = np.max(np.abs(res["mean"][sub] - tm))
    print(f"config-5 mean: |oracle - arbiter| {oracle_err:.3e}, |device - arbiter| {dev_err:.3e}, "
          f"|device - oracle| {np.max(np.abs(res['mean'][sub] - mu)):.3e}, max|y| {np.max(np.abs(y)):.2f}")
    if dev_err > MEAN_RTOL * np.max(np.abs(y)) + MEAN_SLACK * oracle_err:
        problems.append(f"mean: |device - arbiter| {dev_err:.3e} > 1e-5 * {np.max(np.abs(y)):.2f} + 8 * {oracle_err:.3e}")
    acq_t = O.expected_improvement(tm, var, y_best, "min", 0.0) / cost           # the arbiter's mean, the oracle's variance
    oracle_acq_err = np.max(np.abs(acq - acq_t))
    dev_acq_err = np.max(np.abs(res["acq"][sub] - acq_t))
    print(f"config-5 acq: |oracle - arbiter| {oracle_acq_err:.3e}, |device - arbiter| {dev_acq_err:.3e}, max|acq| {amax:.3e}")
    if dev_acq_err > ACQ_TOL_FULL * amax + MEAN_SLACK * oracle_acq_err:
        problems.append(f"acq: |device - arbiter| {dev_acq_err:.3e} > {ACQ_TOL_FULL:.0e} * {amax:.3e} + 8 * {oracle_acq_err:.3e}")
    # the oracle's best among the subsample's candidates is the candidate both device paths chose (or ties it to 1e-4)
    o_best = int(sub[np.argmax(acq[:, 0])])
    o_val = float(np.max(acq[:, 0]))
    if o_best != res["best_idx"] and abs(res["best_val"] - o_val) > BEST_RTOL * abs(o_val):
        problems.append(f"oracle prefers {o_best} ({o_val:.9e}) over the fp32 winner {res['best_idx']} ({res['best_val']:.9e})")
    assert not problems, "; ".join(problems)
    m32.close()
