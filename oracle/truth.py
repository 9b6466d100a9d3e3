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
