"""CPU oracle: numpy/scipy fp64 restatement of the reference's GP posterior + acquisition path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``cbo_with_oop_amd``) never imports anything from ``oracle/`` and has no CPU fallback.

PARITY STATUS: **parity unpinned**.  The arithmetic of this path lives in third-party
packages that are not in /root/reference and are not installable here:
GPy~=1.10.0, emukit~=0.4.10, paramz~=0.9.5 (reference ``requirements.txt:6-7,9``).  The
reference ships no tests, golden vectors or stored outputs for it.
What follows restates (a) the reference's own files line by line where they hold the maths
and (b) the published GPy/emukit algorithms where they do (each function says which).  The
only reference-held known answers are data-level (``tests/golden/*``: the real (X, y)
inputs the reference feeds its GPs and the toy SEM identity); closed-form GP identities pin
the rest (``tests/test_oracle.py``).

All citations ``file:line`` are relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
import scipy.special
import scipy.stats
from scipy.linalg import lapack

# GPy ExactGaussianInference adds this to the likelihood variance on the diagonal
# (GPy 1.10.0 exact_gaussian_inference.py: ``diag.add(Ky, variance + 1e-8)``).
GPY_DIAG_JITTER = 1e-8
# GPy Posterior._raw_predict clips the latent variance here (``np.clip(var, 1e-15, np.inf)``).
GPY_VAR_CLIP = 1e-15
# Hyper-parameters fixed by the reference for CBO-level GPs:
# src/GaussianProcessFactory.py:59-60 (non-causal) and :70-73 (causal).
REF_VARIANCE = 1.0
REF_LENGTHSCALE = 1.0
REF_NOISE_VAR = 1e-10


# --------------------------------------------------------------------------- kernels
def unscaled_sqdist(X, X2, zero_diag=False):
    """GPy ``Stationary._unscaled_dist`` squared (before its sqrt), GPy 1.10.0 stationary.py.

    r2 = -2 X X2^T + (|x|^2 + |x2|^2), clipped at 0.  With ``X2 is None`` GPy also forces
    the diagonal to exactly zero (``zero_diag``); the reference's CausalRBF always passes
    X2 explicitly (src/utils_functions/causal_kernels.py:53-55) so that shortcut is not
    taken on the causal path, while GPy's own RBF (src/GaussianProcessFactory.py:59) takes it.
    """
    X = np.asarray(X, dtype=np.float64)
    X2 = np.asarray(X2, dtype=np.float64)
    X1sq = np.sum(np.square(X), 1)
    X2sq = np.sum(np.square(X2), 1)
    r2 = -2.0 * np.dot(X, X2.T) + (X1sq[:, None] + X2sq[None, :])
    if zero_diag:
        n = min(r2.shape)
        r2[np.arange(n), np.arange(n)] = 0.0
    return np.clip(r2, 0.0, np.inf)


def rbf_K(X, X2, variance=REF_VARIANCE, lengthscale=REF_LENGTHSCALE, zero_diag=False):
    """sigma^2 exp(-0.5 r^2), r = sqrt(r2)/l  (src/utils_functions/causal_kernels.py:55-56, 81-82;
    GPy ``Stationary._scaled_dist`` divides the *unscaled* distance by the lengthscale when not ARD).

    ``lengthscale`` may be a scalar or a length-d vector (ARD: GPy scales the inputs first).
    """
    ls = np.asarray(lengthscale, dtype=np.float64)
    if ls.ndim == 0 or ls.size == 1:
        r = np.sqrt(unscaled_sqdist(X, X2, zero_diag)) / float(ls.reshape(-1)[0])
    else:
        r = np.sqrt(unscaled_sqdist(np.asarray(X) / ls, np.asarray(X2) / ls, zero_diag))
    return variance * np.exp(-0.5 * r ** 2)


def causal_K(X, X2, vX, vX2, variance=REF_VARIANCE, lengthscale=REF_LENGTHSCALE, zero_diag=False):
    """CausalRBF.K: RBF + sqrt(v(X)) sqrt(v(X2))^T  (src/utils_functions/causal_kernels.py:45-62).

    ``vX``/``vX2`` are the already evaluated ``variance_adjustment`` vectors ((n,) or (n,1));
    ``None`` means the non-causal kernel (v == 0).
    """
    K = rbf_K(X, X2, variance, lengthscale, zero_diag)
    if vX is not None:
        a = np.sqrt(np.asarray(vX, dtype=np.float64).reshape(-1, 1))
        b = np.sqrt(np.asarray(vX2, dtype=np.float64).reshape(-1, 1))
        K = K + np.dot(a, b.T)
    return K


def causal_Kdiag(n, vX, variance=REF_VARIANCE):
    """CausalRBF.Kdiag = sigma^2 + v(X)[:,0]  (src/utils_functions/causal_kernels.py:64-79);
    GPy ``Stationary.Kdiag`` = sigma^2 for the non-causal kernel."""
    if vX is None:
        return np.full(n, variance, dtype=np.float64)
    return variance + np.asarray(vX, dtype=np.float64).reshape(-1)


# --------------------------------------------------------------------------- inference
class NotPositiveDefinite(np.linalg.LinAlgError):
    pass


def jitchol(A, maxtries=5):
    """GPy ``util.linalg.jitchol`` (GPy 1.10.0 linalg.py): LAPACK dpotrf(lower); on failure add
    ``mean(diag)*1e-6`` to the diagonal, x10 per retry, at most ``maxtries`` retries.

    Returns (L, jitter_added, n_retries).
    """
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return np.tril(L), 0.0, 0
    diagA = np.diag(A)
    if np.any(diagA <= 0.0):
        raise NotPositiveDefinite("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        L, info = lapack.dpotrf(np.ascontiguousarray(A + np.eye(A.shape[0]) * jitter), lower=1)
        if info == 0:
            return np.tril(L), jitter, num_tries
        jitter *= 10
        num_tries += 1
    raise NotPositiveDefinite("not positive definite, even with jitter.")


class Posterior:
    """What GPy keeps after ``ExactGaussianInference.inference``: woodbury_chol (L),
    woodbury_vector (alpha), plus the inputs needed to predict."""

    def __init__(self, X, y, mX, vX, variance, lengthscale, noise_var, L, alpha, jitter, tries,
                 zero_diag):
        self.X, self.y, self.mX, self.vX = X, y, mX, vX
        self.variance, self.lengthscale, self.noise_var = variance, lengthscale, noise_var
        self.L, self.alpha, self.jitter, self.tries = L, alpha, jitter, tries
        self.zero_diag = zero_diag
        self._woodbury_inv = None

    @property
    def woodbury_inv(self):
        """GPy ``Posterior.woodbury_inv``: dpotri(L) symmetrified."""
        if self._woodbury_inv is None:
            Wi, _ = lapack.dpotri(self.L, lower=1)
            Wi = np.tril(Wi) + np.tril(Wi, -1).T
            self._woodbury_inv = Wi
        return self._woodbury_inv


def fit(X, y, mX=None, vX=None, variance=REF_VARIANCE, lengthscale=REF_LENGTHSCALE,
        noise_var=REF_NOISE_VAR, zero_diag=None):
    """GPy ``ExactGaussianInference.inference`` for the models the reference builds
    (src/GaussianProcessFactory.py:57-73): Ky = K + (noise + 1e-8) I; L = jitchol(Ky);
    alpha = dpotrs(L, y - m(X)).

    ``zero_diag`` defaults to GPy's behaviour: forced-zero diagonal distances for the plain RBF
    (X2=None path), not for CausalRBF (explicit X2).
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    if zero_diag is None:
        zero_diag = vX is None
    K = causal_K(X, X, vX, vX, variance, lengthscale, zero_diag)
    Ky = K.copy()
    Ky[np.diag_indices_from(Ky)] += noise_var + GPY_DIAG_JITTER
    L, jitter, tries = jitchol(Ky)
    resid = y if mX is None else y - np.asarray(mX, dtype=np.float64).reshape(-1, 1)
    alpha, info = lapack.dpotrs(L, resid, lower=1)
    assert info == 0
    return Posterior(X, y, mX, vX, variance, lengthscale, noise_var, L, alpha, jitter, tries,
                     zero_diag)


def predict(post, Xs, mXs=None, vXs=None, include_noise=True, var_form="trtrs"):
    """GPy ``GP.predict`` -> ``Posterior._raw_predict`` (+ mean function, + likelihood variance):
    mu = Kx^T alpha + m(X*);  var = clip(Kdiag - quad, 1e-15) + noise.

    ``var_form``: "trtrs" = sum((L^-1 Kx)^2) (the triangular form BASELINE.json's north_star
    names); "woodbury" = sum((Ky^-1 Kx) * Kx) (GPy's explicit-inverse form).  SURVEY.md §A.2:
    which of the two GPy 1.10.0 uses cannot be verified here; fixtures record their difference.
    Returns (mean (M,1), var (M,1)).
    """
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    Kx = causal_K(post.X, Xs, post.vX, vXs if post.vX is not None else None,
                  post.variance, post.lengthscale, False)
    mu = np.dot(Kx.T, post.alpha)
    Kxx = causal_Kdiag(Xs.shape[0], vXs if post.vX is not None else None, post.variance)
    if var_form == "trtrs":
        V, info = lapack.dtrtrs(post.L, Kx, lower=1)
        assert info == 0
        var = Kxx - np.sum(np.square(V), 0)
    elif var_form == "woodbury":
        var = Kxx - np.sum(np.dot(post.woodbury_inv.T, Kx) * Kx, 0)
    else:
        raise ValueError(var_form)
    var = np.clip(var, GPY_VAR_CLIP, np.inf)[:, None]
    if mXs is not None:
        mu = mu + np.asarray(mXs, dtype=np.float64).reshape(-1, 1)
    if include_noise:
        var = var + post.noise_var
    return mu, var


def log_marginal_likelihood(post):
    """GPy ``ExactGaussianInference.inference``: 0.5*(-n log 2pi - W_logdet - sum(alpha * (Y - m))),
    W_logdet = 2 sum log diag(L)."""
    r = post.y if post.mX is None else post.y - np.asarray(post.mX, dtype=np.float64).reshape(-1, 1)
    n = post.y.shape[0]
    return float(0.5 * (-n * np.log(2 * np.pi) - 2.0 * np.sum(np.log(np.diag(post.L))) - np.sum(post.alpha * r)))


def log_marginal_likelihood_gradients(post):
    """What GPy's ExactGaussianInference hands the optimiser: dL_dK = 0.5 (alpha alpha^T - Ky^-1), contracted with
    dK/dtheta by ``kern.update_gradients_full`` (GPy RBF / Stationary: dK/dvariance = K_rbf / variance,
    dK/dlengthscale_k = K_rbf r_k^2 / lengthscale_k with r_k = (x_ik - x_jk) / lengthscale_k, summed over k when not
    ARD) and dL_dthetaL = trace(dL_dK) for the Gaussian noise variance.  The causal rank-1 term of CausalRBF has no
    parameter of its own, but ``CausalRBF.update_gradients_full`` (src/utils_functions/causal_kernels.py:153-155)
    defers to ``Stationary.update_gradients_full``, whose variance gradient is ``sum(self.K(X, X2) * dL_dK) /
    variance`` with the kernel's OWN K -- for CausalRBF that K includes the rank-1 term (:45-62).  That quirk (it is
    not the derivative of the causal kernel with respect to the RBF variance) is what the reference's optimiser
    follows, so it is restated here; the lengthscale gradient goes through ``dK_dr`` (:84-85), the stationary part
    alone.  Returns (dL/dvariance, dL/dlengthscale (array), dL/dnoise_var)."""
    ls = np.atleast_1d(np.asarray(post.lengthscale, dtype=np.float64))
    dL_dK = 0.5 * (post.alpha @ post.alpha.T - post.woodbury_inv)
    Krbf = rbf_K(post.X, post.X, post.variance, post.lengthscale)
    Kfull = Krbf if post.vX is None else causal_K(post.X, post.X, post.vX, post.vX, post.variance, post.lengthscale)
    d_var = float(np.sum(dL_dK * Kfull) / post.variance)
    diff2 = (post.X[:, None, :] - post.X[None, :, :]) ** 2                       # (N, N, d)
    if ls.size == 1:
        d_ls = np.array([np.sum(dL_dK * Krbf * diff2.sum(-1)) / ls[0] ** 3])
    else:
        d_ls = np.einsum("ij,ij,ijk->k", dL_dK, Krbf, diff2) / ls ** 3
    return d_var, d_ls, float(np.trace(dL_dK))


# paramz ``transformations.Logexp`` (paramz~=0.9.5, /root/reference/requirements.txt:9; not installed: restated from its published
# source): f(x) = where(x > 36, x, log1p(exp(clip(x, -log(DBL_MAX), 36)))), finv(f) = where(f > 36, f, log(expm1(f))),
# gradfactor(f, df) = df * where(f > 36, 1, -expm1(-f)).  GPy constrains RBF.variance, RBF.lengthscale and Gaussian.variance with it.
LOGEXP_LIM = 36.0
LOGEXP_LOG_LIM = float(np.log(np.finfo(np.float64).max))


def logexp_f(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > LOGEXP_LIM, x, np.log1p(np.exp(np.clip(x, -LOGEXP_LOG_LIM, LOGEXP_LIM))))


def logexp_finv(f):
    f = np.asarray(f, dtype=np.float64)
    with np.errstate(over="ignore"):
        return np.where(f > LOGEXP_LIM, f, np.log(np.expm1(f)))


def logexp_gradfactor(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f > LOGEXP_LIM, 1.0, -np.expm1(-f))


def optimize_hyperparameters(X, y, mX=None, vX=None, variance=REF_VARIANCE, lengthscale=REF_LENGTHSCALE,
                             noise_var=REF_NOISE_VAR, fix_noise=False, max_iters=1000, transform="logexp", info=None):
    """GPy ``model.optimize()`` (src/CBO.py:173 through emukit's ``optimize_restarts(1)``: one run from the current
    parameters; src/utils_functions/utils.py:44) as paramz runs it: ``Model.optimize`` -> ``opt_lbfgsb.opt`` =
    ``scipy.optimize.fmin_l_bfgs_b(f_fp, x_init, maxfun=max_iters, maxiter=max_iters)`` with x the Logexp-transformed
    parameters [rbf.variance, rbf.lengthscale(s), Gaussian_noise.variance (unless fixed)], x_init = finv(current values),
    f_fp = ``Model._objective_grads`` = (-log likelihood, -gradient * gradfactor), the model left at x_opt.
    ``transform="log"``: rounds 1-4's parametrisation (theta = exp x).  GPy is not installed: 'parity unpinned'.
    Returns (variance, lengthscale array, noise_var, lml); ``info`` (a dict) receives funcalls / nit / warnflag."""
    from scipy.optimize import fmin_l_bfgs_b
    ls0 = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
    nl = ls0.size
    to_theta = logexp_f if transform == "logexp" else np.exp

    def unpack(x):
        th = to_theta(np.asarray(x, dtype=np.float64))
        return th[0], (th[1] if nl == 1 else th[1:1 + nl]), (noise_var if fix_noise else th[1 + nl])

    def f(x):
        v, l, nz = unpack(x)
        try:
            post = fit(X, y, mX, vX, v, l, nz)
        except np.linalg.LinAlgError:
            return 1e25, np.zeros_like(x)      # (paramz: inf and the last gradient, clipped; finite here for scipy's line search)
        d_var, d_ls, d_noise = log_marginal_likelihood_gradients(post)
        g = np.asarray([d_var, *np.atleast_1d(d_ls)] + ([] if fix_noise else [d_noise]), dtype=np.float64)
        th = to_theta(np.asarray(x, dtype=np.float64))
        g = g * (logexp_gradfactor(th) if transform == "logexp" else th)          # _transform_gradients / d theta / d log theta
        return -log_marginal_likelihood(post), -g

    th0 = np.asarray([variance, *ls0] + ([] if fix_noise else [noise_var]), dtype=np.float64)
    x0 = logexp_finv(th0) if transform == "logexp" else np.log(th0)
    x_opt, _, d = fmin_l_bfgs_b(f, x0, maxfun=int(max_iters), maxiter=int(max_iters))
    if info is not None:
        info.update(funcalls=d["funcalls"], nit=d["nit"], warnflag=d["warnflag"])
    best = x_opt if f(x_opt)[0] < 1e25 else x0
    v, l, nz = unpack(best)
    return float(v), np.atleast_1d(l), float(nz), -f(best)[0]


def predict_gradients(post, Xs, vXs=None):
    """GPy ``GP.predictive_gradients`` as reached through emukit's ``get_prediction_gradients``
    (src/utils_functions/causal_acquisition_functions.py:54): dmu/dx = gradients_X(alpha^T, Xnew, X) and
    dvar/dx = gradients_X(-2 (Ky^-1 Kx)^T, Xnew, X), with Stationary.gradients_X using only the stationary
    part dK/dx* = K_rbf (x_i - x*) / l^2 (the causal rank-1 term and the mean function are not differentiated,
    SURVEY.md §A.2).  Returns (dmean (M,d), dvar (M,d))."""
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    ls = np.atleast_1d(np.asarray(post.lengthscale, dtype=np.float64))
    Krbf = rbf_K(post.X, Xs, post.variance, post.lengthscale)                      # (N, M)
    Kx = causal_K(post.X, Xs, post.vX, vXs if post.vX is not None else None, post.variance, post.lengthscale)
    W, info = lapack.dpotrs(post.L, Kx, lower=1)                                    # Ky^-1 Kx
    diff = (post.X[:, None, :] - Xs[None, :, :]) / (ls ** 2)                        # (N, M, d)
    dmean = np.einsum("n,nm,nmd->md", post.alpha[:, 0], Krbf, diff)
    dvar = -2.0 * np.einsum("nm,nm,nmd->md", W, Krbf, diff)
    return dmean, dvar


def expected_improvement_with_gradients(post, Xs, y_best, mXs=None, vXs=None, task="min", jitter=0.0):
    """CausalExpectedImprovement.evaluate_with_gradients
    (src/utils_functions/causal_acquisition_functions.py:45-67)."""
    mean, variance = predict(post, Xs, mXs, vXs)
    s = np.sqrt(variance)
    dmean_dx, dvariance_dx = predict_gradients(post, Xs, vXs)
    ds_dx = dvariance_dx / (2 * s)
    mean = mean + jitter
    u, pdf, cdf = standard_normal_pdf_cdf(y_best, mean, s)
    imp = s * (u * cdf + pdf)
    dimp = ds_dx * pdf - cdf * dmean_dx
    return (imp, dimp) if task == "min" else (-imp, -dimp)


# --------------------------------------------------------------------------- acquisition
def standard_normal_pdf_cdf(x, mean, standard_deviation):
    """src/utils_functions/causal_acquisition_functions.py:77-88."""
    u = (x - mean) / standard_deviation
    pdf = scipy.stats.norm.pdf(u)
    cdf = scipy.stats.norm.cdf(u)
    return u, pdf, cdf


def expected_improvement(mean, var, y_best, task="min", jitter=0.0):
    """CausalExpectedImprovement.evaluate (src/utils_functions/causal_acquisition_functions.py:27-43).
    task 'max' returns -EI with the *same* u (reference quirk)."""
    s = np.sqrt(var)
    mean = mean + jitter
    u, pdf, cdf = standard_normal_pdf_cdf(y_best, mean, s)
    imp = s * (u * cdf + pdf)
    return imp if task == "min" else -imp


def cost_of_batch(Xs, fix_costs, variable_flags):
    """Cost.evaluate (src/utils_functions/cost_functions.py:11-17) over GraphInterface.cost
    (src/graphs/GraphInterface.py:46-50): sum_i [fix_i + (variable_i ? sum_rows |x[:,i]| : 0)].
    Note the variable part sums |x| over the WHOLE batch column -> a scalar."""
    cost = 0.0
    for i, (f, v) in enumerate(zip(fix_costs, variable_flags)):
        c = f
        if v:
            c = c + np.sum(np.abs(Xs[:, i]))
        cost += c
    return cost


def acquisition_sweep(post, Xs, y_best, mXs=None, vXs=None, task="min", cost=1.0, ei_jitter=0.0,
                      var_form="trtrs"):
    """The batched anchor scoring of src/utils_functions/causal_optimizer.py:52-55, generalised from
    100 random anchors to a deterministic candidate grid: acq = EI / cost on all
    rows, then the best row.  Ties: lowest index wins (``np.argmax``), as CBO.py:275 does across sets.

    Returns (acq (M,1), best_val, best_idx, mean (M,1), var (M,1)).
    """
    mu, var = predict(post, Xs, mXs, vXs, include_noise=True, var_form=var_form)
    acq = expected_improvement(mu, var, y_best, task, ei_jitter) / cost
    idx = int(np.argmax(acq[:, 0]))
    return acq, float(acq[idx, 0]), idx, mu, var


# --------------------------------------------------------------------------- CBO-level selection
def find_current_global(current_y, dict_interventions, task):
    """src/utils_functions/utils.py:8-26."""
    dict_values = {name: [] for name in dict_interventions}
    for variable, value in current_y.items():
        if len(value) > 0:
            dict_values[variable] = np.min(value) if task == "min" else np.max(value)
    opt = min(dict_values, key=dict_values.get) if task == "min" else max(dict_values, key=dict_values.get)
    return dict_values[opt]


def select_next_intervention(acquisition_ys):
    """src/CBO.py:269-277: first index of the maximum."""
    ys = np.asarray([np.asarray(v, dtype=np.float64).reshape(()) for v in acquisition_ys])
    return int(np.where(ys == np.max(ys))[0][0])


# --------------------------------------------------------------------------- do-calculus prior
def do_prior(graph_post, measurements_cols, intervened_index, values, which):
    """DoCalculus.update_do_function / compute_do (src/DoCalculus.py:34-89), intended maths
    (the shipped code crashes on its dict key; this restates what it computes).

    For each candidate row ``values[m]`` build the (N_obs, d_in) intervened inputs: column j of the
    graph GP's inputs is the observed column unless j is intervened (``intervened_index[j] >= 0``),
    in which case it is the constant ``values[m, intervened_index[j]]`` (DoCalculus.py:80-89); predict
    with the graph GP (:77) and average over the N_obs rows (:59-60).  ``which`` 0 -> mean, 1 -> var.
    Returns (M,1).
    """
    obs = np.asarray(measurements_cols, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    out = np.zeros((values.shape[0], 1))
    for m in range(values.shape[0]):
        inp = obs.copy()
        for j, idx in enumerate(intervened_index):
            if idx >= 0:
                inp[:, j] = values[m, idx]
        mu, var = predict(graph_post, inp, include_noise=True)
        out[m, 0] = np.mean((mu, var)[which])
    return out
