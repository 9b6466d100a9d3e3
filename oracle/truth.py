"""ctypes wrapper for oracle/gp_truth_ld.c (extended-precision arbiter; test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgp_truth_ld.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libgp_truth_ld.so"])
        lib = ctypes.CDLL(path)
        P = ctypes.POINTER(ctypes.c_double)
        lib.gp_truth_predict.restype = ctypes.c_int
        lib.gp_truth_predict.argtypes = [ctypes.c_long, ctypes.c_int, P, P, P, P, ctypes.c_double, P,
                                         ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_long,
                                         P, P, P, ctypes.c_int, P, P, P]
        _LIB = lib
    return _LIB


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def truth_predict(X, y, Xs, mX=None, vX=None, mXs=None, vXs=None, variance=1.0, lengthscale=1.0,
                  diag_add=1e-10 + 1e-8, noise_var=1e-10, include_noise=True):
    """Posterior mean/var (M,1) and alpha (N,) computed in 80-bit long double."""
    c = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    X = np.ascontiguousarray(X, dtype=np.float64)
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    n, d = X.shape
    m = Xs.shape[0]
    ls = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64)).copy()
    ard = int(ls.size > 1)
    y_, mX_, vX_, mXs_, vXs_ = c(y), c(mX), c(vX), c(mXs), c(vXs)
    mean = np.empty(m)
    var = np.empty(m)
    alpha = np.empty(n)
    rc = _lib().gp_truth_predict(n, d, _p(X), _p(y_), _p(mX_), _p(vX_), variance, _p(ls), ard, diag_add,
                                 noise_var, m, _p(Xs), _p(mXs_), _p(vXs_), int(include_noise),
                                 _p(mean), _p(var), _p(alpha))
    if rc != 0:
        raise np.linalg.LinAlgError(f"truth cholesky failed at pivot {rc}")
    return mean[:, None], var[:, None], alpha


def _exact_rbf(X, X2, variance, lengthscale):
    """sigma^2 exp(-r^2/2) with r^2 from DIRECT coordinate differences (accurate to a few eps; GPy's
    |x|^2 + |x'|^2 - 2 x.x' loses eps * |x|^2 absolutely, 1e-9 at the coral graph's T ~ 2475)."""
    ls = np.broadcast_to(np.atleast_1d(np.asarray(lengthscale, dtype=np.float64)), (X.shape[1],))
    r2 = np.zeros((X.shape[0], X2.shape[0]))
    for k in range(X.shape[1]):
        d = (X[:, k][:, None] - X2[:, k][None, :]) / ls[k]
        d *= d
        r2 += d
    r2 *= -0.5
    np.exp(r2, out=r2)
    r2 *= variance
    return r2


def refined_mean(post, Xs, mXs=None, vXs=None, iterations=4, exact_entries=False):
    """Posterior mean K*^T alpha + m(X*) with alpha from ITERATIVE REFINEMENT of Ky alpha = y - m(X): an fp64 Cholesky
    factor is the solver, the residuals are formed in long double (x87 80-bit) against the matrix entries.  Converges
    while eps * cond(Ky) < 1; the refined alpha carries a relative error of a few eps, so the mean no longer carries the
    eps * cond(Ky) * |k*| |alpha| of a plain fp64 solve.  ``exact_entries=False``: the very entries the oracle's fit
    factored (GPy's distance formula) -- arbitrates the SOLVE.  ``exact_entries=True``: entries from direct coordinate
    differences, refactored with the fit's diagonal add and jitter -- the mean of the exact-arithmetic kernel on the same
    inputs, what the all-long-double restatement (gp_truth_predict, O(n^3) scalar work) computes; the arbiter at sizes
    where that one is not affordable (plain models only: no causal rank-1 term).  Test infrastructure only (tests/)."""
    from scipy.linalg import lapack
    from . import gp_oracle as O
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    if exact_entries:
        assert post.vX is None, "exact_entries: plain RBF models only"
        K = _exact_rbf(post.X, post.X, post.variance, post.lengthscale)
        K[np.diag_indices_from(K)] += post.noise_var + O.GPY_DIAG_JITTER + post.jitter
        L, info = lapack.dpotrf(K, lower=1, clean=0, overwrite_a=0)
        assert info == 0, info
        Kx = _exact_rbf(post.X, Xs, post.variance, post.lengthscale)
    else:
        K = O.causal_K(post.X, post.X, post.vX, post.vX, post.variance, post.lengthscale, post.zero_diag)
        K[np.diag_indices_from(K)] += post.noise_var + O.GPY_DIAG_JITTER + post.jitter
        L = post.L
        Kx = O.causal_K(post.X, Xs, post.vX, vXs, post.variance, post.lengthscale, False)
    r = post.y if post.mX is None else post.y - np.asarray(post.mX, dtype=np.float64).reshape(-1, 1)
    rl = r.astype(np.longdouble)
    a0, info = lapack.dpotrs(L, r, lower=1)
    assert info == 0
    alpha = a0.astype(np.longdouble)
    block = 2048                                            # long-double copies of K one row block at a time
    for _ in range(iterations):
        res = np.empty_like(rl)
        for i in range(0, K.shape[0], block):
            res[i:i + block] = rl[i:i + block] - K[i:i + block].astype(np.longdouble) @ alpha
        d, info = lapack.dpotrs(L, np.asarray(res, dtype=np.float64), lower=1)
        assert info == 0
        alpha = alpha + d.astype(np.longdouble)
    mean = np.asarray(Kx.astype(np.longdouble).T @ alpha, dtype=np.float64)
    if mXs is not None:
        mean = mean + np.asarray(mXs, dtype=np.float64).reshape(-1, 1)
    return mean, np.asarray(alpha, dtype=np.float64)


def refined_variance(post, Xs, vXs=None, iterations=4, include_noise=True):
    """Posterior variance at a FEW points for the exact-arithmetic kernel on the oracle's inputs (entries from direct
    coordinate differences, the fit's diagonal add and jitter): kss - k*^T Ky^-1 k* with Ky w = k* solved by iterative
    refinement (fp64 Cholesky factor as the solver, long-double residuals) and the final inner product in long double.
    The arbiter of the variance at sizes where the all-long-double restatement (gp_truth_predict) is not affordable;
    O(n^2 m) long-double work per iteration, so keep m small (tens).  Plain RBF models only.  Test infrastructure only."""
    from scipy.linalg import lapack
    from . import gp_oracle as O
    assert post.vX is None, "plain RBF models only"
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    K = _exact_rbf(post.X, post.X, post.variance, post.lengthscale)
    K[np.diag_indices_from(K)] += post.noise_var + O.GPY_DIAG_JITTER + post.jitter
    L, info = lapack.dpotrf(K, lower=1, clean=0, overwrite_a=0)
    assert info == 0, info
    Kx = _exact_rbf(post.X, Xs, post.variance, post.lengthscale)
    kl = Kx.astype(np.longdouble)
    w0, info = lapack.dpotrs(L, Kx, lower=1)
    assert info == 0
    w = w0.astype(np.longdouble)
    block = 1024
    for _ in range(iterations):
        res = np.empty_like(kl)
        for i in range(0, K.shape[0], block):
            res[i:i + block] = kl[i:i + block] - K[i:i + block].astype(np.longdouble) @ w
        d, info = lapack.dpotrs(L, np.asarray(res, dtype=np.float64), lower=1)
        assert info == 0
        w = w + d.astype(np.longdouble)
    var = np.longdouble(post.variance) - np.sum(kl * w, axis=0)
    var = np.maximum(var, np.longdouble(O.GPY_VAR_CLIP))
    if include_noise:
        var = var + np.longdouble(post.noise_var)
    return np.asarray(var, dtype=np.float64)[:, None]
