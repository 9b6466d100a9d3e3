"""Acquisition optimiser of the MI355X path, mirroring the class name and ``optimize`` contract of
/root/reference/src/utils_functions/causal_optimizer.py:15-71.

The reference samples ``num_anchor_points`` uniform random anchors, scores them with one batched
``acquisition.evaluate`` (:52-55), keeps the best and refines it with L-BFGS (:59-65).  Here the batched
scoring IS the optimiser: a deterministic regular grid over the space (BASELINE.json's "N-candidate sweep",
SURVEY.md §0.7), scored on the GPU, arg-max taken on the GPU.  Gradient refinement is SURVEY.md §8 f3.
"""
from __future__ import annotations

import numpy as np

from ..graphs import meshgrid_candidates
from .causal_acquisition_functions import CandidateGrid


class CausalGradientAcquisitionOptimizer:
    def __init__(self, space, num_anchor_points=None, grid_shape=None):
        """``space``: emukit ParameterSpace / object with ``parameters`` / list of (lo, hi).  ``grid_shape``
        (points per dimension) defaults to ``default_grid_shape``; ``num_anchor_points``, if given, is used as
        the candidate budget instead of 16384."""
        from .utils import default_grid_shape, space_bounds
        self.space = space
        self.bounds = space_bounds(space)
        self.grid_shape = list(grid_shape) if grid_shape is not None else default_grid_shape(
            len(self.bounds), budget=num_anchor_points)
        self._grid, self._grid_model = None, None

    def candidates(self):
        return meshgrid_candidates(self.bounds, self.grid_shape)

    def refine(self, acquisition, x0):
        """The reference's second stage (causal_optimizer.py:59-65; emukit OptLbfgs = scipy fmin_l_bfgs_b with
        bounds and maxfun=1000, minimising -acquisition with its gradient) started from ``x0`` ((1,d))."""
        from scipy.optimize import fmin_l_bfgs_b

        def f_df(v):
            f, df = acquisition.evaluate_with_gradients(v[None, :])
            return -float(f[0, 0]), -df[0]

        x, fx, _ = fmin_l_bfgs_b(f_df, np.asarray(x0, dtype=np.float64).reshape(-1), bounds=self.bounds, maxfun=1000)
        return x[None, :], np.array([[-fx]])

    def optimize(self, acquisition, context=None, refine=False):
        """(x_max (1,d), acquisition value at x_max (1,1)) -- emukit ``AcquisitionOptimizerBase.optimize``.
        ``acquisition`` is ``CausalExpectedImprovement(...) / Cost(...)`` (an ``AcquisitionQuotient``) or a bare
        ``CausalExpectedImprovement``."""
        # the grid of this optimiser stays on the device while the model object is the same (no allocation per call)
        model = acquisition.model
        if self._grid is None or self._grid_model is not model:
            if self._grid is not None:
                self._grid.close()
            self._grid, self._grid_model = CandidateGrid(self.candidates(), model), model
        grid, pts = self._grid, self._grid.points
        res = acquisition.sweep(grid)
        x = pts[res["best_idx"]][None, :].copy()
        fx = np.array([[res["best_val"]]])
        if refine:
            xr, fr = self.refine(acquisition, x)
            if fr[0, 0] >= fx[0, 0]:
                x, fx = xr, fr
        return x, fx
