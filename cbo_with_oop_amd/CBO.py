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


class _FixedCosts:
    """Batch costs decided by the caller (the whole grid's, when a rank only holds a block of it)."""

    def __init__(self, by_set, order):
        self.values = [by_set[s] for s in order]


class CBOAcquisitionPath:
    """Holds exactly the state those methods read on the reference's ``CBO`` object: ``gp_type``,
    ``exploration_set``, ``costs``, ``task``, per-set data, spaces, prior closures and models."""

    def __init__(self, gp_type, exploration_set, costs, task, data_x, data_y, space_list, mean_functions=None,
                 var_functions=None, grid_shapes=None, keep_solutions=True, comm="env"):
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
        self._grid_points = {}    # per set: (space object, grid shape object, shape, points)
        self._call_cache = {}     # handle arrays and batch costs of the multi-set sweep, valid while the objects are
        # Several GPUs (one process per GPU): ``comm`` is a sharding.Communicator (or anything with world / rank /
        # argmax), "env" = the launcher's (RANK / WORLD_SIZE), None = single process.  Whole exploration sets go to
        # ranks when there are at least as many sets as ranks (S = 25 coral sets on 8 GPUs: every GPU sweeps full
        # grids of three sets), candidate blocks of every set otherwise; either way one 16-byte exchange per set.
        self._comm = comm
        # keep L^-1 K* of every set's grid on the device: a trial then costs the set intervened on one forward
        # solve and one new row (append-only step) instead of a refit and a full sweep
        self.keep_solutions = bool(keep_solutions)

    def update_all_gaussian_processes(self):
        """CBO.py:209-222."""
        self._call_cache.clear()          # its handle arrays name the models and grids replaced below
        for _, grid in self._grids.values():
            grid.close()
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
        self._call_cache.clear()
        old = self._grids.pop(s, None)
        if old is not None:
            old[1].close()
        self.models[s] = GPFactory.create(self.gp_type, self.data_x[s], self.data_y[s],
                                          [self.mean_functions[s], self.var_functions[s]], emukit_wrapper=True, fit=fit)

    @property
    def comm(self):
        if isinstance(self._comm, str):                      # "env": formed on first use (touches the GPU)
            from .sharding import default_communicator
            self._comm = default_communicator()
        return self._comm

    def placement(self):
        """("single" | "sets" | "candidates", world, rank)."""
        comm = self.comm
        if comm is None or comm.world == 1:
            return "single", 1, 0
        return ("sets" if self.es_size >= comm.world else "candidates"), comm.world, comm.rank

    def grid_points(self, s):
        """(shape, points) of set s's regular grid; built once per (space object, grid shape) -- replace
        ``space_list[s]`` / ``grid_shapes[s]`` to change it (the cache keeps the space object alive, so its identity
        cannot be reused)."""
        space, want = self.space_list[s], self.grid_shapes[s]
        hit = self._grid_points.get(s)
        if hit is not None and hit[0] is space and hit[1] is want:
            return hit[2], hit[3]
        bounds = space_bounds(space)
        shape = tuple(want or default_grid_shape(len(bounds)))
        pts = meshgrid_candidates(bounds, shape)
        self._grid_points[s] = (space, want, shape, pts)
        return shape, pts

    def candidate_grid(self, s):
        """The regular grid over the set's box, resident on the device across trials (the reference draws fresh
        random anchors each trial; the grid of BASELINE.json's sweep is fixed).  Rebuilt when the grid shape, the
        model object or the prior closures change.  Under candidate placement it is this rank's contiguous block of
        the grid (``index_offset`` = the block's first row); ``full_points`` keeps the whole grid for the look-up of
        the winner."""
        mode, world, rank = self.placement()
        shape, pts = self.grid_points(s)
        key = (shape, id(self.models[s]), id(self.mean_functions[s]), id(self.var_functions[s]), mode, world, rank)
        cached = self._grids.get(s)
        if cached is None or cached[0] != key:
            if cached is not None:
                self._call_cache.clear()
                cached[1].close()
            if mode == "candidates":
                from .sharding import shard_bounds
                begin, end = shard_bounds(pts.shape[0], world, rank)
                grid = CandidateGrid(pts[begin:max(end, begin + 1)] if end > begin else pts[:1], self.models[s],
                                     index_offset=begin, keep_solution=self.keep_solutions)
                grid.empty_shard = end <= begin
            else:
                grid = CandidateGrid(pts, self.models[s], keep_solution=self.keep_solutions)
                grid.empty_shard = False
            grid.full_points = pts
            cached = (key, grid)
            self._grids[s] = cached
        return cached[1]

    def compute_best_acquisition_values(self, current_best):
        """CBO.py:237-260: the loop over the exploration sets, as ONE device call (``cbo_acq_sweep_sets``); across
        several GPUs, this rank's share of it and one arg-max exchange per set."""
        mode, world, rank = self.placement()
        if mode == "single":
            grids = [self.candidate_grid(s) for s in range(self.es_size)]
            return find_next_y_points(self.models, current_best, self.exploration_set, self.costs, self.task, grids,
                                      cache=self._call_cache)
        from .sharding import ERROR_CANDIDATE, NO_CANDIDATE
        from .utils_functions.cost_functions import Cost
        # A rank that fails (a model that is not positive definite, a device error) must not leave the others blocked in
        # an exchange: every rank takes part in one failure-flag exchange per trial and all of them raise when it is set.
        local, failure = {}, None
        try:
            mine = [s for s in range(self.es_size) if mode == "candidates" or s % world == rank]
            mine = [s for s in mine if not (mode == "candidates" and self.candidate_grid(s).empty_shard)]
            if mine:
                grids = [self.candidate_grid(s) for s in mine]
                # batch costs are those of the WHOLE grid (a variable cost sums |x| over the batch column)
                full_cost = {s: float(Cost(self.costs, self.exploration_set[s]).evaluate(self.candidate_grid(s).full_points))
                             for s in mine}
                _, ys = find_next_y_points([self.models[s] for s in mine], current_best,
                                           [self.exploration_set[s] for s in mine], _FixedCosts(full_cost, mine), self.task,
                                           grids, cache=self._call_cache, raw=True)
                local = {s: ys[i] for i, s in enumerate(mine)}
        except Exception as exc:  # noqa: BLE001 -- re-raised below, after the exchanges
            failure = exc
        # One explicit flag per trial says whether any rank failed (cbo_comm_max_f64 of 0 / 1): it is not encoded in the
        # arg-max payload, where a healthy rank's genuine NaN acquisition (NaN is maximal, lowest index wins) could beat an
        # error record and leave the ranks disagreeing about whether to go on.  Every rank raises when it is set, before any
        # arg-max exchange.  (A communicator without ``max`` -- a caller's own object with only world / rank / argmax --
        # keeps the error record in the payload.)
        flag = getattr(self.comm, "max", None)
        winners, failed_somewhere = [], False
        if flag is not None:
            failed_somewhere = float(flag(1.0 if failure is not None else 0.0)) > 0.0
            if failure is not None:
                raise failure
            if failed_somewhere:
                raise RuntimeError("compute_best_acquisition_values: another rank failed during this trial's sweep")
        for s in range(self.es_size):
            val, idx = (float("nan"), ERROR_CANDIDATE) if failure is not None else local.get(s, (-np.inf, NO_CANDIDATE))
            val, idx = self.comm.argmax(val, idx)                       # identical on every rank
            failed_somewhere = failed_somewhere or idx == ERROR_CANDIDATE
            winners.append((val, idx))
        if failure is not None:
            raise failure
        if failed_somewhere:
            raise RuntimeError("compute_best_acquisition_values: another rank failed during this trial's sweep")
        xs, out = [], []
        for s in range(self.es_size):
            val, idx = winners[s]
            pts = self.grid_points(s)[1]
            x_new = pts[idx][None, :].copy()
            cost = Cost(self.costs, self.exploration_set[s])
            batch, point = float(cost.evaluate(pts)), float(cost.evaluate(x_new))
            # utils.py:36 re-evaluates EI / cost at x_new alone: the same EI over the point's own cost
            out.append(np.array([[val if point == batch else val * batch / point]]))
            xs.append(x_new)
        return xs, out

    def trial_step(self, current_best):
        """One trial of the reference's loop between two observations (src/CBO.py:143-173, ``CBO.intervene``) as ONE
        device-library call: ``update_gaussian_process_of_last_intervention`` (the model of the set intervened on last takes
        ``data_x / data_y`` of that set), ``compute_best_acquisition_values`` and ``select_next_intervention``.  Returns
        ``(xs, ys, (exploration set, index))`` -- what those three return -- and leaves ``last_intervention`` at the pick.
        At the reference's model sizes the three calls' host glue costs as much as the one launch that serves them
        (``cbo_trial_step``); anything the one call does not cover (several ranks, a model rebuilt with other prior closures
        or other hyper-parameters, the first trial) takes the three calls."""
        import ctypes
        from . import _lib
        from .utils_functions.utils import winners_to_points
        s = self.last_intervention
        st = self._call_cache.get("sweep_sets")
        model = self.models[s] if (s is not None and self.models) else None
        fast = (st is not None and model is not None and (self.comm is None or self.comm.world == 1)
                and model.mean_function is self.mean_functions[s] and model.variance_adjustment is self.var_functions[s]
                and model._hyper_initial and st["cost_table"] is self.costs
                and st["models"] == self.models)             # (lists of the same objects: compared by identity first)
        if fast:
            grids, cached = st["grids"], self._grids
            for i in range(self.es_size):
                entry = cached.get(i)
                if entry is None or entry[1] is not grids[i]:
                    fast = False
                    break
        if not fast:
            if s is not None:
                self.update_gaussian_process_of_last_intervention()
            xs, ys = self.compute_best_acquisition_values(current_best)
            return xs, ys, self.select_next_intervention(ys)
        model._set_arrays(self.data_x[s], self.data_y[s])
        pm, pv = model._prior(model.X)
        st["y_best"].fill(current_best if type(current_best) is float else
                          float(np.asarray(current_best, dtype=np.float64).reshape(-1)[0]))
        fixed = st.get("trial_args")             # the pointers that do not change from trial to trial, made once
        if fixed is None:
            chosen = ctypes.c_int(-1)
            fixed = st["trial_args"] = (_lib.load().cbo_trial_step, _lib.dptr(st["y_best"]), _lib.dptr(st["batch_cost"]),
                                        _lib.dptr(st["vals"]), st["idxs"].ctypes.data_as(_lib.c_int64_p), chosen,
                                        ctypes.byref(chosen))
        call, y_best, batch_cost, vals, idxs, chosen, chosen_ref = fixed
        rc = call(self.es_size, st["gps"], st["cds"], s, model.X.shape[0], _lib.dptr(model.X), _lib.dptr(model._y_flat),
                  _lib.dptr(pm), _lib.dptr(pv), y_best, _lib.TASK_CODE[self.task], 0.0, batch_cost, vals, idxs, chosen_ref)
        if rc:
            _lib.check(rc)
        model.stale = model.small            # (a larger model was refitted by the general path inside the call)
        for m in self.models:
            if not m.small:
                m.stale = False
        xs, ys = winners_to_points(st, self.models, st["grids"], current_best, self.task)
        self.last_intervention = chosen.value
        return xs, ys, (self.exploration_set[chosen.value], chosen.value)

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
