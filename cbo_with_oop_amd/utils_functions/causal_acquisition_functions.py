"""Causal Expected Improvement on the MI355X path.

Mirrors /root/reference/src/utils_functions/causal_acquisition_functions.py:8-43 (constructor and
``evaluate``); the arithmetic runs in kernels_acq.hip behind ``cbo_acq_sweep`` (include/cbo_hip.h).
``CandidateGrid`` is the device-resident candidate set that replaces the 100 random anchors of
causal_optimizer.py:52.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib
from ..GaussianProcessFactory import _column


class CandidateGrid:
    """Candidate interventions (M,d) resident in HBM (cbo_cands).  ``index_offset`` makes reported
    arg-max indices global when this is one shard of a larger grid (one shard per GPU)."""

    def __init__(self, points, model=None, index_offset=0, context=None, keep_solution=False):
        self.points = _lib.as_f64(points)
        if self.points.ndim != 2:
            raise ValueError("points must be (M, d)")
        self._lib = _lib.load()
        ctx = context if context is not None else (model._ctx if model is not None else _lib.Context.get())
        self._ctx = ctx
        pm = pv = None
        if model is not None and model.causal:
            m = self.points.shape[0]
            pm = _column(model.mean_function(self.points), m, "mean_function")
            pv = _column(model.variance_adjustment(self.points), m, "variance_adjustment")
        self.index_offset = int(index_offset)
        self._handle = ctypes.c_void_p()
        _lib.check(self._lib.cbo_cands_create(ctx.handle, self.points.shape[0], self.points.shape[1],
                                              _lib.dptr(self.points), _lib.dptr(pm), _lib.dptr(pv),
                                              self.index_offset, ctypes.byref(self._handle)))
        if keep_solution:
            # V = L^-1 K* stays on the device after a sweep, so that a model extended by ``append`` costs this
            # candidate set one new row instead of the whole substitution (n_pad * m doubles of HBM)
            _lib.check(self._lib.cbo_cands_keep_solution(self._handle, 1))

    def __len__(self):
        return self.points.shape[0]

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            if not self._ctx.closed:
                self._lib.cbo_cands_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CausalExpectedImprovement:
    def __init__(self, current_global_min, task, model, jitter=0.0):
        """Same signature as the reference (:10-25)."""
        self.model = model
        self.jitter = jitter
        self.current_global_min = current_global_min
        self.task = task

    def sweep(self, candidates, cost=1.0, want_acq=False, want_posterior=False, refit=False):
        """Score every candidate and pick the best: returns dict(best_val, best_idx, acq, mean, var).
        ``candidates`` is a CandidateGrid (device resident) or an (M,d) array.  ``refit=True`` refits the model
        from its resident data first, overlapped with the sweep (``cbo_gp_fit_sweep``): the pair of steps
        ``CBO.intervene`` takes for the set it has just intervened on.  A model whose data were replaced with
        ``set_data(..., fit=False)`` is refitted this way without being asked."""
        own = not isinstance(candidates, CandidateGrid)
        grid = CandidateGrid(candidates, self.model) if own else candidates
        m = len(grid)
        acq = np.empty(m) if want_acq else None
        mean = np.empty(m) if want_posterior else None
        var = np.empty(m) if want_posterior else None
        best_val = ctypes.c_double(0.0)
        best_idx = ctypes.c_int64(-1)
        args = (self.model._handle, grid._handle, float(np.asarray(self.current_global_min).reshape(-1)[0]),
                _lib.TASK_CODE[self.task], float(self.jitter), float(cost), _lib.dptr(acq), _lib.dptr(mean),
                _lib.dptr(var), ctypes.byref(best_val), ctypes.byref(best_idx))
        try:
            if refit or self.model.stale:
                tries, jitter = ctypes.c_int(0), ctypes.c_double(0.0)
                _lib.check(_lib.load().cbo_gp_fit_sweep(*args, ctypes.byref(tries), ctypes.byref(jitter)))
                self.model._note_jitter(tries.value, jitter.value)
            else:
                _lib.check(_lib.load().cbo_acq_sweep(*args))
        finally:
            if own:
                grid.close()
        col = lambda a: None if a is None else a[:, None]
        return {"best_val": best_val.value, "best_idx": best_idx.value, "acq": col(acq), "mean": col(mean),
                "var": col(var)}

    def evaluate(self, x):
        """(M,1) improvement, as the reference's ``evaluate`` (:27-43)."""
        return self.sweep(x, cost=1.0, want_acq=True)["acq"]

    @property
    def has_gradients(self):
        return True

    def evaluate_with_gradients(self, x):
        """(improvement (M,1), d improvement / d x (M,d)) as the reference's ``evaluate_with_gradients``
        (:45-67).  Posterior and its gradients come from the device; the closing arithmetic on M (a few) points is
        host numpy, scipy's normal pdf/cdf as in the reference (:77-88)."""
        import scipy.stats
        x = _lib.as_f64(x)
        mean, variance = self.model.predict(x)
        standard_deviation = np.sqrt(variance)
        dmean_dx, dvariance_dx = self.model.get_prediction_gradients(x)
        dstandard_deviation_dx = dvariance_dx / (2 * standard_deviation)
        mean = mean + self.jitter
        u = (np.asarray(self.current_global_min, dtype=np.float64).reshape(-1)[0] - mean) / standard_deviation
        pdf, cdf = scipy.stats.norm.pdf(u), scipy.stats.norm.cdf(u)
        improvement = standard_deviation * (u * cdf + pdf)
        dimprovement_dx = dstandard_deviation_dx * pdf - cdf * dmean_dx
        if self.task == 'min':
            return improvement, dimprovement_dx
        return -improvement, -dimprovement_dx

    def __truediv__(self, cost):
        """``CausalExpectedImprovement(...) / Cost(...)`` as in src/utils_functions/utils.py:34 (emukit's
        ``Acquisition.__truediv__`` builds a Quotient there)."""
        return AcquisitionQuotient(self, cost)


class AcquisitionQuotient:
    """emukit ``Quotient`` of the improvement and a ``Cost``: ``evaluate(x) = EI(x) / Cost(x)`` where the cost of
    a batch is one scalar (cost_functions.py module docstring); the division happens inside the HIP sweep."""

    def __init__(self, numerator, denominator):
        self.numerator, self.denominator = numerator, denominator
        self.model = numerator.model

    def _points(self, candidates):
        return candidates.points if isinstance(candidates, CandidateGrid) else np.asarray(candidates)

    def sweep(self, candidates, **kwargs):
        return self.numerator.sweep(candidates, cost=float(self.denominator.evaluate(self._points(candidates))),
                                    **kwargs)

    def evaluate(self, x):
        return self.sweep(x, want_acq=True)["acq"]

    def evaluate_with_gradients(self, x):
        """Quotient rule with the reference's zero cost gradient (cost_functions.py:23-24)."""
        x = _lib.as_f64(x)
        f, df = self.numerator.evaluate_with_gradients(x)
        c = float(self.denominator.evaluate(x))
        return f / c, df / c

    @property
    def has_gradients(self):
        return True
