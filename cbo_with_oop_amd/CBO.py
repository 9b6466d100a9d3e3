"""The four methods of the reference's agent that drive the hot path, as a mixin/standalone class
(/root/reference/src/CBO.py:209-277).  The experiment loop, monitor, do-calculus and graph classes stay
the reference's own (out of scope); INTEGRATION.md shows the two-line change that makes
``src/CBO.py`` use this module.
"""
from __future__ import annotations

import numpy as np

from .GaussianProcessFactory import GaussianProcessFactory as GPFactory
from .graphs import meshgrid_candidates
from .utils_functions.causal_acquisition_functions import CandidateGrid
from .utils_functions.utils import (default_grid_shape, find_current_global, find_next_y_point, find_next_y_points,  # noqa: F401
                                    space_bounds)


class CBOAcquisitionPath:
    """Holds exactly the state those methods read on the reference's ``CBO`` object: ``gp_type``,
    ``exploration_set``, ``costs``, ``task``, per-set data, spaces, prior closures and models."""

    def __init__(self, gp_type, exploration_set, costs, task, data_x, data_y, space_list, mean_functions=None,
                 var_functions=None, grid_shapes=None, keep_solutions=True):
        self.gp_type = gp_type
        self.exploration_set = exploration_set
        self.es_size = len(exploration_set)
        self.costs = costs
        self.task = task
        self.data_x, self.data_y = data_x, data_y
        self.space_list = space_list
        self.mean_functions = mean_functions or [None] * self.es_size
        self.var_functions = var_functions or [None] * self.es_size
        self.grid_shapes = grid_shapes or [None] * self.es_size
        self.intervention_names = ["".join(v) for v in exploration_set]
        self.models = []
        self.last_intervention = None
        self._grids = {}          # per set: (grid shape, prior closures, device-resident candidate grid)
        self._call_cache = {}     # handle arrays and batch costs of the multi-set sweep, valid while the objects are
        # keep L^-1 K* of every set's grid on the device: a trial then costs the set intervened on one forward
        # solve and one new row (append-only step) instead of a refit and a full sweep
        self.keep_solutions = bool(keep_solutions)

    def update_all_gaussian_processes(self):
        """CBO.py:209-222."""
        self._grids.clear()
        self.models = [
            GPFactory.create(self.gp_type, self.data_x[s], self.data_y[s],
                             [self.mean_functions[s], self.var_functions[s]], emukit_wrapper=True)
            for s in range(self.es_size)
        ]

    def update_gaussian_process_of_last_intervention(self, fit=False):
        """CBO.py:224-235.  By default the rebuilt model is left unfitted: ``compute_best_acquisition_values``
        comes next (CBO.py:152-164) and its sweep over this set refits and sweeps in one overlapped device call.
        ``fit=True`` restores the reference's timing (a not-PD error then surfaces here)."""
        s = self.last_intervention
        model = self.models[s]
        if model is not None and model.mean_function is self.mean_functions[s] \
                and model.variance_adjustment is self.var_functions[s]:
            model.rebuild(self.data_x[s], self.data_y[s], fit=fit)      # same handle: no allocation, no new grid
            return
        self._grids.pop(s, None)
        self.models[s] = GPFactory.create(self.gp_type, self.data_x[s], self.data_y[s],
                                          [self.mean_functions[s], self.var_functions[s]], emukit_wrapper=True, fit=fit)

    def candidate_grid(self, s):
        """The regular grid over the set's box, resident on the device across trials (the reference draws fresh
        random anchors each trial; the grid of BASELINE.json's sweep is fixed).  Rebuilt when the grid shape, the
        model object or the prior closures change."""
        bounds = space_bounds(self.space_list[s])
        shape = tuple(self.grid_shapes[s] or default_grid_shape(len(bounds)))
        key = (shape, id(self.models[s]), id(self.mean_functions[s]), id(self.var_functions[s]))
        cached = self._grids.get(s)
        if cached is None or cached[0] != key:
            if cached is not None:
                cached[1].close()
            cached = (key, CandidateGrid(meshgrid_candidates(bounds, shape), self.models[s],
                                         keep_solution=self.keep_solutions))
            self._grids[s] = cached
        return cached[1]

    def compute_best_acquisition_values(self, current_best):
        """CBO.py:237-260: the loop over the exploration sets, as ONE device call (``cbo_acq_sweep_sets``)."""
        grids = [self.candidate_grid(s) for s in range(self.es_size)]
        return find_next_y_points(self.models, current_best, self.exploration_set, self.costs, self.task, grids,
                                  cache=self._call_cache)

    def current_best_solution(self, current_best_y):
        """CBO.py:262-267 (the monitor's ``current_best_y`` dict is passed in)."""
        return find_current_global(current_best_y, self.intervention_names, self.task)

    def select_next_intervention(self, acquisition_ys):
        """CBO.py:269-277: first index of the maximum."""
        ys = np.asarray([np.asarray(y, dtype=np.float64).reshape(-1)[0] for y in acquisition_ys])
        indices = int(np.where(ys == np.max(ys))[0][0])
        self.last_intervention = indices
        return self.exploration_set[indices], self.last_intervention

    # BASELINE.json's north_star calls it select_intervention
    select_intervention = select_next_intervention
