"""``find_current_global`` and ``find_next_y_point`` of /root/reference/src/utils_functions/utils.py:8-37
for the MI355X path.  ``find_next_y_point`` keeps the reference signature; the acquisition optimiser
behind it is a dense candidate-grid sweep on the GPU instead of 100 random anchors + L-BFGS
(BASELINE.json configs).
"""
from __future__ import annotations

import os

import numpy as np

from ..graphs import meshgrid_candidates
from .causal_acquisition_functions import CandidateGrid, CausalExpectedImprovement
from .cost_functions import Cost


def find_current_global(current_y, dict_interventions, task):
    """utils.py:8-26: best value observed so far over all exploration sets."""
    dict_values = {}
    for j in range(len(dict_interventions)):
        dict_values[dict_interventions[j]] = []
    for variable, value in current_y.items():
        if len(value) > 0:
            if task == 'min':
                dict_values[variable] = np.min(current_y[variable])
            else:
                dict_values[variable] = np.max(current_y[variable])
    if task == 'min':
        opt_variable = min(dict_values, key=dict_values.get)
    else:
        opt_variable = max(dict_values, key=dict_values.get)
    return dict_values[opt_variable]


def fit_gaussian_process(x, y, parameter_list):
    """utils.py:40-45: graph-level GP, RBF(lengthscale=p[0], variance=p[1], ARD=p[3]), noise fixed 1e-2.
    ``gp.optimize()`` (hyper-parameter MLE) is called as in the reference."""
    from ..GaussianProcessFactory import GaussianProcessFactory, GaussianProcessType
    gp = GaussianProcessFactory.create(GaussianProcessType.GRAPH_GP, x, y, parameter_list)
    gp.optimize()
    return gp


def space_bounds(space):
    """[(lo, hi)] from an emukit ParameterSpace (``get_bounds()``), from objects with ``.parameters``
    carrying ``.min/.max`` (graph_functions.py:80-93 builds ContinuousParameter(name, min, max)), or
    from a plain list of pairs."""
    if hasattr(space, "get_bounds"):
        return [tuple(b) for b in space.get_bounds()]
    if hasattr(space, "parameters"):
        return [(p.min, p.max) for p in space.parameters]
    return [tuple(b) for b in space]


def default_grid_shape(d, budget=None):
    """Points per dimension for a d-dimensional sweep.  BASELINE.json: 200 per 1-D toy set, and the
    16k grid 32x32x16 for d=3; other d get the largest equal split within the same 16k budget."""
    if budget is None:
        budget = int(os.environ.get("CBO_HIP_GRID_BUDGET", "16384"))
    if d == 1:
        return [min(budget, 200)]
    if d == 3 and budget == 16384:
        return [32, 32, 16]
    n = max(2, int(np.floor(budget ** (1.0 / d))))
    return [n] * d


def find_next_y_point(space, model, current_global_best, evaluated_set, costs_functions, task='min',
                      grid_shape=None, candidates=None, anchors="grid", num_anchor_points=None):
    """utils.py:29-37.  Returns (y_acquisition (1,1), x_new (1,d)).

    ``candidates`` (optional (M,d) array or CandidateGrid) overrides the regular grid over ``space``.
    ``anchors="uniform"``: the reference's own optimiser instead of the grid -- 100 uniform anchors from numpy's global
    generator, one batched device sweep over them, L-BFGS from the best (causal_optimizer.py:26-65), then the
    acquisition re-evaluated at the point found (utils.py:36) -- the four lines of the reference's function.
    """
    cost_acquisition = Cost(costs_functions, evaluated_set)
    if anchors == "uniform":
        from .causal_optimizer import CausalGradientAcquisitionOptimizer
        optimizer = CausalGradientAcquisitionOptimizer(space, num_anchor_points=num_anchor_points, anchors="uniform")
        acquisition = CausalExpectedImprovement(current_global_best, task, model) / cost_acquisition
        x_new, _ = optimizer.optimize(acquisition)
        return acquisition.evaluate(x_new), x_new
    ei = CausalExpectedImprovement(current_global_best, task, model)
    own = False
    if candidates is None:
        bounds = space_bounds(space)
        pts = meshgrid_candidates(bounds, grid_shape or default_grid_shape(len(bounds)))
        grid, own = CandidateGrid(pts, model), True
    elif isinstance(candidates, CandidateGrid):
        grid = candidates
    else:
        grid, own = CandidateGrid(candidates, model), True
    try:
        batch_cost = float(cost_acquisition.evaluate(grid.points))      # ONE scalar for the batch (Quotient)
        res = ei.sweep(grid, cost=batch_cost)
        x_new = grid.points[res["best_idx"] - grid.index_offset][None, :].copy()
        # utils.py:36 re-evaluates the acquisition at x_new alone; only variable costs change the value
        point_cost = float(cost_acquisition.evaluate(x_new))
        if point_cost == batch_cost:
            y = np.array([[res["best_val"]]])
        else:
            y = ei.sweep(x_new, cost=point_cost, want_acq=True)["acq"]
    finally:
        if own:
            grid.close()
    return y, x_new


def find_next_y_points(models, current_global_best, evaluated_sets, costs_functions, task, grids, cache=None, raw=False):
    """``find_next_y_point`` for every exploration set of a trial in ONE device call (``cbo_acq_sweep_sets``): the loop
    of src/CBO.py:249-257.  ``grids[s]`` is the CandidateGrid of set s.  Models with at most 128 observations -- all
    the reference builds -- are factored and swept inside one launch and need not be fitted; the others go through
    the general path inside the same call.  ``cache`` (a dict the caller keeps between trials) holds what does not
    change while models, grids and cost functions stay the same objects: the handle arrays and the batch costs.
    Returns (xs, ys): lists of (1,d) points and (1,1) acquisition values.  ``raw=True`` (the multi-GPU caller): the
    batch costs are given (``costs_functions.values``, those of the whole grid) and ys are (value, global index)
    pairs for the arg-max exchange, xs is None."""
    import ctypes
    from .. import _lib
    s = len(models)
    # The cache entry holds the models, grids and cost table themselves (strong references, compared with ``is``)
    # next to their device handles: an ``id()`` alone comes back as soon as CPython reuses a freed address, and the
    # handle array would then name destroyed cbo_gp / cbo_cands objects.
    handles = tuple(int(o._handle.value or 0) for o in list(models) + list(grids))      # ctypes.c_void_p handles
    st = cache.get("sweep_sets") if cache is not None else None
    same = (st is not None and not raw and st["cost_table"] is costs_functions and st["handles"] == handles
            and len(st["models"]) == s and all(a is b for a, b in zip(st["models"], models))
            and all(a is b for a, b in zip(st["grids"], grids)))
    if not same:
        if 0 in handles:
            raise ValueError("find_next_y_points: a model or candidate grid has been closed")
        costs = None if raw else [Cost(costs_functions, evaluated_sets[i]) for i in range(s)]
        st = {"cost_table": costs_functions, "models": list(models), "grids": list(grids), "handles": handles,
              "costs": costs,
              "batch_cost": np.array(costs_functions.values if raw else
                                     [float(costs[i].evaluate(grids[i].points)) for i in range(s)], dtype=np.float64),
              "gps": (ctypes.c_void_p * s)(*[m._handle for m in models]),
              "cds": (ctypes.c_void_p * s)(*[g._handle for g in grids]),
              "y_best": np.empty(s), "vals": np.empty(s), "idxs": np.empty(s, dtype=np.int64)}
        if cache is not None:
            cache["sweep_sets"] = st
    costs, batch_cost, vals, idxs = st["costs"], st["batch_cost"], st["vals"], st["idxs"]
    st["y_best"][:] = float(np.asarray(current_global_best, dtype=np.float64).reshape(-1)[0])
    _lib.check(_lib.load().cbo_acq_sweep_sets(s, st["gps"], st["cds"], _lib.dptr(st["y_best"]), _lib.TASK_CODE[task], 0.0,
                                              _lib.dptr(batch_cost), _lib.dptr(vals),
                                              idxs.ctypes.data_as(_lib.c_int64_p)))
    for i in range(s):
        if not models[i].small:                  # the general path fitted it on the way (deferred refit)
            models[i].stale = False
    if raw:
        return None, [(float(vals[i]), int(idxs[i])) for i in range(s)]
    return winners_to_points(st, models, grids, current_global_best, task)


def winners_to_points(st, models, grids, current_global_best, task):
    """(xs, ys) of utils.py:36 from the winners a multi-set sweep left in ``st`` (the cache entry of
    ``find_next_y_points``): the grid point of every set and its acquisition value re-evaluated at that point alone --
    only variable costs change the value."""
    costs, batch_cost, vals, idxs = st["costs"], st["batch_cost"], st["vals"], st["idxs"]
    xs, ys = [], []
    winners, values, batch = idxs.tolist(), vals.tolist(), batch_cost.tolist()
    for i in range(len(models)):
        j = winners[i] - grids[i].index_offset
        x_new = grids[i].points[j:j + 1].copy()
        point_cost = float(costs[i].evaluate(x_new))
        if point_cost == batch[i]:
            y = np.array(((values[i],),))
        else:
            y = CausalExpectedImprovement(current_global_best, task, models[i]).sweep(x_new, cost=point_cost,
                                                                                      want_acq=True)["acq"]
        xs.append(x_new)
        ys.append(y)
    return xs, ys
