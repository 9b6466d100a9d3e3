"""Host-side mirror of the reference's GP factory for the MI355X path.

Same names, argument meaning and error behaviour as /root/reference/src/GaussianProcessFactory.py:9-73;
the object ``create`` returns answers the calls the reference makes on a GPy ``GPRegression`` and on
emukit's ``GPyModelWrapper`` around it, but every number comes from the HIP kernels
behind libcbo_hip.so (include/cbo_hip.h).  There is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from enum import IntEnum

import numpy as np

from . import _lib


class GaussianProcessType(IntEnum):
    """src/GaussianProcessFactory.py:9-15."""
    GRAPH_GP = 0
    CAUSAL_GP = 1
    NON_CAUSAL_GP = 2


# paramz ``transformations.Logexp`` (paramz~=0.9.5, requirements.txt:9), the constraint GPy puts on every positive parameter
# of these models (``RBF.variance`` / ``.lengthscale``, ``Gaussian.variance``): the optimiser works on x, the model sees
# theta = log(1 + exp(x)).  Restated here because GPy's ``model.optimize()`` (src/CBO.py:173, src/utils_functions/utils.py:44)
# runs scipy's L-BFGS-B in THAT space, from x0 = finv(theta0): same routine + same space + same start = its trajectory.
_LOGEXP_LIM = 36.0
_LOGEXP_LOG_LIM = float(np.log(np.finfo(np.float64).max))


def logexp_f(x):
    """Logexp.f: theta(x)."""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > _LOGEXP_LIM, x, np.log1p(np.exp(np.clip(x, -_LOGEXP_LOG_LIM, _LOGEXP_LIM))))


def logexp_finv(theta):
    """Logexp.finv: x(theta)."""
    theta = np.asarray(theta, dtype=np.float64)
    with np.errstate(over="ignore"):
        return np.where(theta > _LOGEXP_LIM, theta, np.log(np.expm1(theta)))


def logexp_gradfactor(theta):
    """Logexp.gradfactor without the df factor: d theta / d x = 1 - exp(-theta)."""
    theta = np.asarray(theta, dtype=np.float64)
    return np.where(theta > _LOGEXP_LIM, 1.0, -np.expm1(-theta))


def _column(values, n, what):
    """A reference closure returns (k,1) (DoCalculus.py:66); accept (k,), (k,1) or a scalar."""
    v = np.asarray(values, dtype=np.float64)
    if v.ndim == 0:
        v = np.full(n, float(v))
    v = np.ascontiguousarray(v.reshape(-1))
    if v.shape[0] != n:
        raise ValueError(f"{what} returned {v.shape[0]} values for {n} points")
    return v


class HipGaussianProcess:
    """GP posterior resident on one MI355X.  Duck-types the two objects the reference uses:

    * GPy ``GPRegression``: ``predict(Xnew)`` -> (mean (M,1), var (M,1)) with the Gaussian likelihood
      noise included (used by src/DoCalculus.py:77), ``X``, ``Y``, ``set_XY``, ``optimize``.
    * emukit ``GPyModelWrapper``: ``predict``, ``set_data`` (src/Monitor.py:160), ``optimize``
      (src/CBO.py:173), ``X``, ``Y``, ``model``.
    """

    def __init__(self, x, y, *, variance=1.0, lengthscale=1.0, ard=False, noise_var=1e-10, mean_function=None,
                 variance_adjustment=None, zero_diag=None, context=None, fix_noise=False, fit=True, dtype=None):
        if (mean_function is None) != (variance_adjustment is None):
            raise ValueError("mean_function and variance_adjustment must be given together")
        self._lib = _lib.load()
        self._ctx = context if context is not None else _lib.Context.get()
        # "f64" (default) or "f32": the fit is fp64 either way; "f32" runs every sweep / predict of this model on the
        # f32 MFMA from a once-per-fit fp32 copy of the factor (BASELINE.json configs[4]; include/cbo_hip.h)
        self.dtype = dtype if dtype is not None else os.environ.get("CBO_HIP_DTYPE", "f64")
        if self.dtype not in _lib.DTYPE_CODE:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPE_CODE)}")
        self.mean_function = mean_function
        self.variance_adjustment = variance_adjustment
        self.causal = mean_function is not None
        self.variance = float(variance)
        self.noise_var = float(noise_var)
        self.ard = bool(ard)
        self.fix_noise = bool(fix_noise)
        x = _lib.as_f64(x)
        if x.ndim != 2:
            raise ValueError("x must be (N, d)")
        self.input_dim = x.shape[1]
        ls = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
        if self.ard and ls.size == 1:
            ls = np.full(self.input_dim, ls[0])
        self.lengthscale = np.ascontiguousarray(ls)
        # GPy's plain RBF takes the X2=None shortcut (zero diagonal distance); CausalRBF passes X2
        # explicitly (causal_kernels.py:53-55)
        self.zero_diag = (not self.causal) if zero_diag is None else bool(zero_diag)
        self._handle = ctypes.c_void_p()
        self._set_arrays(x, y)
        pm, pv = self._prior(self.X)
        _lib.check(self._lib.cbo_gp_create(
            self._ctx.handle, _lib.DTYPE_CODE[self.dtype], self.X.shape[0], self.input_dim, _lib.dptr(self.X), _lib.dptr(self._y_flat),
            _lib.dptr(pm), _lib.dptr(pv), self.variance, _lib.dptr(self.lengthscale), int(self.ard), self.noise_var,
            int(self.zero_diag), ctypes.byref(self._handle)))
        self._initial_hyper = (self.variance, self.lengthscale.copy(), self.noise_var)
        self._hyper_initial = True
        if fit:
            self._fit()
        else:
            self.stale = True      # the first acquisition sweep refits, overlapped (see set_data(fit=False))

    # -- construction helpers ------------------------------------------------------------------
    def _set_arrays(self, x, y):
        x = _lib.as_f64(x)
        y = _lib.as_f64(y)
        if y.ndim == 1:
            y = y[:, None]
        if x.ndim != 2 or y.shape != (x.shape[0], 1):
            raise ValueError(f"expected x (N,d) and y (N,1), got {x.shape} and {y.shape}")
        if x.shape[1] != self.input_dim:
            raise ValueError("input dimension changed")
        self.X, self.Y = x, y
        self._y_flat = np.ascontiguousarray(y[:, 0])

    def _prior(self, pts):
        if not self.causal:
            return None, None
        n = pts.shape[0]
        return (_column(self.mean_function(pts), n, "mean_function"),
                _column(self.variance_adjustment(pts), n, "variance_adjustment"))

    def _fit(self):
        tries = ctypes.c_int(0)
        jitter = ctypes.c_double(0.0)
        _lib.check(self._lib.cbo_gp_fit(self._handle, ctypes.byref(tries), ctypes.byref(jitter)))
        self._note_jitter(tries.value, jitter.value)

    def fit_level(self, level):
        """ONE level of jitchol's ladder (``cbo_gp_fit_level``; sharding.fit_over_ranks walks the ladder with one level
        per rank): ``(outcome, jitter)``, outcome 1 = factored (the model is fitted with ``jitter_tries = level``), 0 =
        not positive definite at this level, -1 = non-positive diagonal entries."""
        status = ctypes.c_int(0)
        jitter = ctypes.c_double(0.0)
        _lib.check(self._lib.cbo_gp_fit_level(self._handle, int(level), ctypes.byref(status), ctypes.byref(jitter)))
        if status.value == 1:
            self._note_jitter(int(level), jitter.value)
        return status.value, jitter.value

    def adopted_factor(self, level):
        """The factor at ``level`` has arrived from another rank (``cbo_comm_share_factor``): the host-side state follows."""
        tries = ctypes.c_int(0)
        jitter = ctypes.c_double(0.0)
        _lib.check(self._lib.cbo_gp_jitter(self._handle, ctypes.byref(tries), ctypes.byref(jitter)))
        self._note_jitter(tries.value, jitter.value)

    def take_factor_slices(self, source, level, n_owners):
        """The receiving side of ``cbo_comm_share_factor`` with device copies in the place of the transfers
        (``cbo_gp_take_factor_slices``): this model takes ``source``'s factor at ``level`` in the ``n_owners`` row slices
        the owners would send, and adopts it.  For tests on one GPU."""
        _lib.check(self._lib.cbo_gp_take_factor_slices(self._handle, source._handle, int(level), int(n_owners)))
        self.adopted_factor(level)
        self.stale = False

    stale = False      # data uploaded, posterior not yet refitted (set_data(..., fit=False))

    @property
    def small(self):
        """At most 128 observations on the fp64 path: the multi-set sweep (``cbo_acq_sweep_sets``) factors and sweeps
        such a model inside one launch, from its resident data, whether it is fitted or not."""
        return self.dtype == "f64" and self.X.shape[0] <= 128

    def _note_jitter(self, tries, jitter):
        self.stale = False
        self.jitter_tries, self.jitter = tries, jitter
        if tries:
            # GPy logs a warning when jitchol needed jitter
            warnings.warn(f"Added jitter of {jitter:.10e}", RuntimeWarning, stacklevel=4)

    # -- reference-facing API -------------------------------------------------------------------
    @property
    def model(self):
        """emukit's wrapper exposes the GPy model as ``.model``; here they are the same object."""
        return self

    def predict(self, x, include_likelihood=True):
        """(mean (M,1), var (M,1)); GP.predict / GPyModelWrapper.predict."""
        x = _lib.as_f64(x)
        if x.ndim != 2 or x.shape[1] != self.input_dim:
            raise ValueError(f"x must be (M, {self.input_dim})")
        m = x.shape[0]
        pm, pv = self._prior(x)
        mean = np.empty(m)
        var = np.empty(m)
        self.ensure_fitted()
        _lib.check(self._lib.cbo_gp_predict(self._handle, m, _lib.dptr(x), _lib.dptr(pm), _lib.dptr(pv),
                                            int(include_likelihood), _lib.dptr(mean), _lib.dptr(var)))
        return mean[:, None], var[:, None]

    def predict_grouped(self, x, group, include_likelihood=True):
        """Predict at ``x`` ((M*group, d)) and average mean and variance over each consecutive run of ``group``
        rows on the device: (mean (M,1), var (M,1)).  This is the do-calculus reduction of
        src/DoCalculus.py:59-60 (np.mean over the observed rows of one intervention)."""
        x = _lib.as_f64(x)
        if x.ndim != 2 or x.shape[1] != self.input_dim or x.shape[0] % group:
            raise ValueError(f"x must be (M*group, {self.input_dim})")
        m = x.shape[0] // group
        pm, pv = self._prior(x)
        mean = np.empty(m)
        var = np.empty(m)
        self.ensure_fitted()
        _lib.check(self._lib.cbo_gp_predict_grouped(self._handle, m, group, _lib.dptr(x), _lib.dptr(pm), _lib.dptr(pv),
                                                    int(include_likelihood), _lib.dptr(mean), _lib.dptr(var)))
        return mean[:, None], var[:, None]

    def predict_do(self, observed, intervened_index, values, include_likelihood=True):
        """The do-calculus reduction with the inputs built on the device (``cbo_gp_predict_do``): for every row of
        ``values`` ((M, n_iv)) predict at the rows of ``observed`` ((N_obs, d)) with column j replaced by
        ``values[:, intervened_index[j]]`` where ``intervened_index[j] >= 0``, and average over the rows
        (src/DoCalculus.py:59-60, 74-89).  Returns (mean (M,1), var (M,1)); nothing of size M * N_obs exists on the host."""
        observed = _lib.as_f64(observed)
        values = _lib.as_f64(values)
        if values.ndim == 1:
            values = values[None, :]
        if observed.ndim != 2 or observed.shape[1] != self.input_dim:
            raise ValueError(f"observed must be (N_obs, {self.input_dim})")
        idx = np.ascontiguousarray(intervened_index, dtype=np.int32)
        if idx.shape != (self.input_dim,):
            raise ValueError("intervened_index needs one entry per input column")
        m = values.shape[0]
        mean, var = np.empty(m), np.empty(m)
        self.ensure_fitted()
        _lib.check(self._lib.cbo_gp_predict_do(self._handle, m, observed.shape[0], _lib.dptr(observed), values.shape[1],
                                               _lib.dptr(values), idx.ctypes.data_as(_lib.c_int_p), int(include_likelihood),
                                               _lib.dptr(mean), _lib.dptr(var)))
        return mean[:, None], var[:, None]

    def predict_noiseless(self, x):
        return self.predict(x, include_likelihood=False)

    def set_data(self, X, Y, fit=True):
        """GPyModelWrapper.set_data -> GP.set_XY: replace the data and refit (src/Monitor.py:160).  When the new data
        are the resident ones plus one observation -- what that call site passes every trial -- the factor grows by
        one column on the device (``append``) instead of being rebuilt; same posterior up to rounding.

        ``fit=False`` uploads only and leaves the refit to the next use: an acquisition sweep then runs the
        refit and the sweep overlapped (``cbo_gp_fit_sweep``); any other consumer (predict, log_likelihood, ...)
        fits first.  A not-positive-definite error then surfaces at that use instead of here."""
        # (a model of at most 128 observations is refactored inside every multi-set sweep anyway: no append there)
        if not (self.dtype == "f64" and np.shape(X)[0] <= 128 and not fit) and self._grew_by_one_row(X, Y) \
                and self.append(np.asarray(X)[-1], np.asarray(Y).reshape(-1)[-1]):
            return                                   # the factor grew by one column instead of being rebuilt
        self._set_arrays(X, Y)
        pm, pv = self._prior(self.X)
        if not fit:
            _lib.check(self._lib.cbo_gp_upload_data(self._handle, self.X.shape[0], _lib.dptr(self.X),
                                                    _lib.dptr(self._y_flat), _lib.dptr(pm), _lib.dptr(pv)))
            self.stale = True
            return
        _lib.check(self._lib.cbo_gp_set_data(self._handle, self.X.shape[0], _lib.dptr(self.X),
                                             _lib.dptr(self._y_flat), _lib.dptr(pm), _lib.dptr(pv)))
        self.stale = False
        tries = ctypes.c_int(0)
        jitter = ctypes.c_double(0.0)
        _lib.check(self._lib.cbo_gp_jitter(self._handle, ctypes.byref(tries), ctypes.byref(jitter)))
        self.jitter_tries, self.jitter = tries.value, jitter.value

    def append(self, x_new, y_new):
        """One more observation (what every CBO trial adds to the set it intervened on): the factor grows by one
        column on the device instead of being rebuilt (``cbo_gp_append``).  Returns False -- and changes nothing --
        when the shortcut does not apply (model not fitted yet, jitter in the factor, padded size exhausted,
        non-positive pivot); the caller then uses ``set_data``."""
        if self.stale:
            return False
        x_new = _lib.as_f64(x_new).reshape(1, self.input_dim)
        y_val = float(np.asarray(y_new, dtype=np.float64).reshape(-1)[0])
        pm, pv = self._prior(x_new)
        done = ctypes.c_int(0)
        _lib.check(self._lib.cbo_gp_append(self._handle, _lib.dptr(x_new), y_val, float(pm[0]) if pm is not None else 0.0,
                                           float(pv[0]) if pv is not None else 0.0, ctypes.byref(done)))
        if not done.value:
            return False
        self.X = np.vstack([self.X, x_new])
        self.Y = np.vstack([self.Y, [[y_val]]])
        self._y_flat = np.ascontiguousarray(self.Y[:, 0])
        return True

    def _grew_by_one_row(self, X, Y):
        """The new data are the resident ones plus one observation (what src/Monitor.py:148-160 produces every
        trial) and the model is fitted: the append shortcut applies."""
        if self.stale:
            return False
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64).reshape(-1, 1)
        return (X.ndim == 2 and X.shape[0] == self.X.shape[0] + 1 and X.shape[1] == self.X.shape[1]
                and Y.shape[0] == X.shape[0] and np.array_equal(X[:-1], self.X) and np.array_equal(Y[:-1], self.Y))

    def ensure_fitted(self):
        """Fit now if the data were replaced with ``set_data(..., fit=False)`` and nothing has refitted since."""
        if self.stale:
            self._fit()

    set_XY = set_data

    def log_likelihood(self):
        """GPy ``model.log_likelihood()``: log marginal likelihood of the fitted model (device reduction over
        the factor's diagonal and z = L^-1 (y - m))."""
        out = ctypes.c_double(0.0)
        self.ensure_fitted()
        _lib.check(self._lib.cbo_gp_log_marginal(self._handle, ctypes.byref(out)))
        return out.value

    def set_hyperparameters(self, variance, lengthscale, noise_var, fit=True):
        """Replace kernel variance, lengthscale(s) and Gaussian noise variance and refit (``fit=False``: the next
        use refits, as with ``set_data``)."""
        ls = np.atleast_1d(np.asarray(lengthscale, dtype=np.float64))
        if self.ard and ls.size == 1:
            ls = np.full(self.input_dim, ls[0])
        ls = np.ascontiguousarray(ls)
        _lib.check(self._lib.cbo_gp_set_hyper(self._handle, float(variance), _lib.dptr(ls), float(noise_var)))
        self.variance, self.lengthscale, self.noise_var = float(variance), ls, float(noise_var)
        v0, ls0, nv0 = self._initial_hyper
        self._hyper_initial = (self.variance, self.noise_var) == (v0, nv0) and np.array_equal(ls, ls0)
        self.stale = True            # the device model is unfitted from here on; _fit clears the flag when it succeeds
        if fit:
            self._fit()

    def hyper_is_initial(self):
        """The hyper-parameters are the ones the constructor was given (what a model rebuilt by the reference starts
        from); they only change through ``set_hyperparameters``, which keeps the answer."""
        return self._hyper_initial

    def rebuild(self, X, Y, fit=True):
        """What the reference obtains by constructing a NEW model on new data (src/CBO.py:224-235 builds one each
        trial): the hyper-parameters the constructor was given, the new data.  Same device handle and buffers, so
        nothing is allocated when the padded size does not change."""
        if not self.hyper_is_initial():
            v0, ls0, nv0 = self._initial_hyper
            self.set_hyperparameters(v0, ls0, nv0, fit=False)
        self.set_data(X, Y, fit=fit)                 # appends when the data merely grew by one row

    def log_likelihood_gradients(self):
        """(d log p(y)/d variance, d/d lengthscale (array), d/d noise_var) of the fitted model: the gradients GPy's
        inference hands its optimiser, computed on the device (``cbo_gp_lml_gradients``).  A model of at most 128
        observations is not fitted for this: one launch goes from the data and the current hyper-parameters to the
        likelihood and its gradients (the general path, jitchol ladder included, takes over when that fails)."""
        if not self.small:
            self.ensure_fitted()
        lml, dv, dn = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        dls = np.zeros(self.lengthscale.size)
        _lib.check(self._lib.cbo_gp_lml_gradients(self._handle, ctypes.byref(lml), ctypes.byref(dv), _lib.dptr(dls),
                                                  ctypes.byref(dn)))
        self._last_lml = lml.value
        return dv.value, dls, dn.value

    def _objective(self, x, transform="log"):
        """(negative log marginal likelihood, its gradient with respect to x) at theta(x) = [variance, lengthscale(s),
        (noise)]: ``transform="logexp"``: theta = log(1 + exp(x)), paramz's ``Model._objective_grads`` with
        ``_transform_gradients`` (gradient times 1 - exp(-theta)); ``"log"``: theta = exp(x)."""
        x = np.asarray(x, dtype=np.float64)
        theta = logexp_f(x) if transform == "logexp" else np.exp(x)
        nl = self.lengthscale.size
        noise = self.noise_var if self.fix_noise else theta[1 + nl]
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                self.set_hyperparameters(theta[0], theta[1:1 + nl], noise, fit=not self.small)
            dv, dls, dn = self.log_likelihood_gradients()
        except np.linalg.LinAlgError:
            # paramz hands the optimiser inf and the clipped gradient of the last good point (and gives up after ten such
            # evaluations in a row); a large finite value keeps scipy's line search defined and is rejected the same way
            return 1e25, np.zeros_like(x)
        g = np.asarray([dv, *dls] + ([] if self.fix_noise else [dn]), dtype=np.float64)
        g = g * (logexp_gradfactor(theta) if transform == "logexp" else theta)
        return -self._last_lml, -g

    def optimize(self, max_iters=1000, transform="logexp", **kwargs):
        """Hyper-parameter MLE (emukit ``GPyModelWrapper.optimize`` -> GPy ``optimize_restarts(1, robust=True)`` -- one run from
        the current parameters, nothing drawn --, src/CBO.py:173; ``gp.optimize()`` in src/utils_functions/utils.py:44):
        maximise the log marginal likelihood over kernel variance, lengthscale(s) and -- unless fixed, as for graph-level
        GPs -- the noise variance.  Host logic, as paramz has it (``Model.optimize`` -> ``opt_lbfgsb.opt``):
        ``scipy.optimize.fmin_l_bfgs_b(f_fp, x0, maxfun=max_iters, maxiter=max_iters)`` on the Logexp-transformed
        parameters from ``x0 = finv(theta0)``, the model left at the routine's ``x_opt``.  Every evaluation is a device
        refit plus the device likelihood and its analytic gradients.  ``transform="log"`` is rounds 1-4's parametrisation
        (theta = exp(x), same stationary point, another path to it).  GPy is not installed here: the trajectory is GPy's
        by construction, not by comparison (parity unpinned)."""
        from scipy.optimize import OptimizeResult, fmin_l_bfgs_b
        if transform not in ("logexp", "log"):
            raise ValueError("transform must be 'logexp' (GPy / paramz) or 'log'")
        theta0 = [self.variance, *self.lengthscale]
        if not self.fix_noise:
            theta0.append(self.noise_var)
        theta0 = np.asarray(theta0, dtype=np.float64)
        x0 = logexp_finv(theta0) if transform == "logexp" else np.log(theta0)
        x_opt, f_opt, info = fmin_l_bfgs_b(lambda x: self._objective(x, transform), x0, maxfun=int(max_iters),
                                           maxiter=int(max_iters))
        f_opt = self._objective(x_opt, transform)[0]          # opt_lbfgsb: f_opt = f_fp(x_opt)[0]; the model sits at x_opt
        if f_opt >= 1e25:                     # the factorisation failed at the point the optimiser settled on
            nl = self.lengthscale.size        # (GPy restores the previous parameters after a failed step): back to the
            self.set_hyperparameters(theta0[0], theta0[1:1 + nl],      # starting point, which was fitted before
                                     self.noise_var if self.fix_noise else theta0[1 + nl])
        res = OptimizeResult(x=x_opt, fun=f_opt, jac=info["grad"], nfev=info["funcalls"], nit=info["nit"],
                             status=info["warnflag"], message=info["task"], success=info["warnflag"] == 0,
                             transform=transform)
        self.optimization_result = res
        return res

    def get_prediction_gradients(self, x):
        """emukit ``GPyModelWrapper.get_prediction_gradients`` -> GPy ``predictive_gradients``:
        (d mean / d x (M,d), d var / d x (M,d)).  As in GPy, the mean function's and the causal rank-1 term's own
        gradients are not included.  Any number of points (a whole grid included): forward and
        backward substitution of the batch on the device."""
        x = _lib.as_f64(x)
        if x.ndim != 2 or x.shape[1] != self.input_dim:
            raise ValueError(f"x must be (M, {self.input_dim})")
        m = x.shape[0]
        pv = _column(self.variance_adjustment(x), m, "variance_adjustment") if self.causal else None
        dmean = np.empty((m, self.input_dim))
        dvar = np.empty((m, self.input_dim))
        self.ensure_fitted()
        _lib.check(self._lib.cbo_gp_predict_gradients(self._handle, m, _lib.dptr(x), _lib.dptr(pv), _lib.dptr(dmean),
                                                      _lib.dptr(dvar)))
        return dmean, dvar

    # -- posterior state (GPy: model.posterior.woodbury_chol / woodbury_vector) -------------------
    def posterior_state(self):
        n = self.X.shape[0]
        L = np.empty((n, n))
        alpha = np.empty(n)
        self.ensure_fitted()
        _lib.check(self._lib.cbo_gp_get_posterior(self._handle, _lib.dptr(L), _lib.dptr(alpha)))
        return L, alpha[:, None]

    def assembled_Ky(self):
        n = self.X.shape[0]
        K = np.empty((n, n))
        _lib.check(self._lib.cbo_gp_assemble_kxx(self._handle, _lib.dptr(K)))
        return K

    # -- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            if not self._ctx.closed:               # after cbo_shutdown the device memory is gone with the context
                self._lib.cbo_gp_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GaussianProcessFactory:
    """src/GaussianProcessFactory.py:18-73, same static methods."""

    @staticmethod
    def create(gp_type, x, y, parameters=None, emukit_wrapper=False, fit=True, dtype=None):
        """``fit=False`` (not in the reference) builds the model without fitting it; the first acquisition sweep
        then refits and sweeps in one overlapped call.  ``dtype="f32"`` (not in the reference either; default from
        ``CBO_HIP_DTYPE``, else "f64") selects the fp32 sweep."""
        gp_functions = {
            GaussianProcessType.GRAPH_GP: GaussianProcessFactory.create_graph_gp,
            GaussianProcessType.CAUSAL_GP: GaussianProcessFactory.create_causal_gp,
            GaussianProcessType.NON_CAUSAL_GP: GaussianProcessFactory.create_non_causal_gp,
        }
        # emukit_wrapper only selected the wrapper class in the reference; HipGaussianProcess answers
        # both interfaces, so the flag changes nothing here.
        return gp_functions[gp_type](x, y, parameters, fit=fit, dtype=dtype)

    @staticmethod
    def create_graph_gp(x, y, parameters, fit=True, dtype=None):
        """:49-54  RBF(lengthscale=p[0], variance=p[1], ARD=p[3]), noise fixed to 1e-2 after construction."""
        return HipGaussianProcess(x, y, variance=parameters[1], lengthscale=parameters[0], ard=parameters[3],
                                  noise_var=1e-2, fix_noise=True, fit=fit, dtype=dtype)

    @staticmethod
    def create_non_causal_gp(x, y, _, fit=True, dtype=None):
        """:57-60  RBF(lengthscale=1, variance=1), noise 1e-10."""
        return HipGaussianProcess(x, y, variance=1.0, lengthscale=1.0, noise_var=1e-10, fit=fit, dtype=dtype)

    @staticmethod
    def create_causal_gp(x, y, parameters, fit=True, dtype=None):
        """:63-73  CausalRBF(variance_adjustment=var_function) + mean function, noise 1e-10."""
        mean_function, var_function = parameters
        return HipGaussianProcess(x, y, variance=1.0, lengthscale=1.0, noise_var=1e-10,
                                  mean_function=mean_function, variance_adjustment=var_function, fit=fit, dtype=dtype)
