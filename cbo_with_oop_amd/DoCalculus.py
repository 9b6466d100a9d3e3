"""Do-calculus prior on the MI355X path: the mean / variance closures the reference
feeds to ``create_causal_gp`` (/root/reference/src/DoCalculus.py:14-89).

For every candidate intervention value the reference builds the observed inputs of a graph-level GP with
the intervened columns overwritten (``get_intervened_inputs``, :80-89), predicts with that GP (:77) and
averages over the observed rows (:59-60) -- a Python loop over candidates with a dict cache.  Here all
candidates go through ONE batched device predict whose per-candidate row means are reduced on the GPU
(``cbo_gp_predict_do``: the intervened inputs are expanded on the device too).  The shipped reference indexes a dict with a list and passes the raw
interventions record instead of the variable names, so this restates the
intended computation, with the exploration-set variable names as the "intervention".
"""
from __future__ import annotations

from functools import partial

import numpy as np


def intervened_inputs(observed, intervened_index, values):
    """(M*N_obs, d_in) inputs: ``observed`` (N_obs, d_in) tiled per candidate, column j replaced by
    ``values[m, intervened_index[j]]`` where ``intervened_index[j] >= 0`` (DoCalculus.py:74-76, 80-89)."""
    observed = np.asarray(observed, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    m, n_obs = values.shape[0], observed.shape[0]
    out = np.broadcast_to(observed[None, :, :], (m, n_obs, observed.shape[1])).copy()
    for j, idx in enumerate(intervened_index):
        if idx >= 0:
            out[:, :, j] = values[:, idx][:, None]
    return out.reshape(m * n_obs, observed.shape[1])


def do_function(gp, observed, intervened_index, index, values):
    """``update_do_function`` for all rows of ``values`` at once: (M,1) mean (index 0) or variance (index 1)
    of the graph GP's predictive distribution averaged over the observed rows."""
    values = np.asarray(values, dtype=np.float64)
    if values.ndim == 1:
        values = values[None, :]
    # the (M * N_obs, d) intervened inputs are built on the device from `observed` and `values`
    mean, var = gp.predict_do(observed, intervened_index, values)
    return np.float64(mean if index == 0 else var)


def do_prior_functions(gp, observed, intervened_index):
    """(mean_function, variance_function) for ``GaussianProcessFactory.create_causal_gp``."""
    return (partial(do_function, gp, observed, intervened_index, 0),
            partial(do_function, gp, observed, intervened_index, 1))


class DoCalculus:
    """Same method names as the reference class.  ``cbo`` must expose ``exploration_set``, ``es_size``,
    ``measurements`` (mapping variable name -> column) and ``graph`` with ``get_gp_name`` and
    ``fit_dependencies`` (as src/graphs/GraphInterface.py and the graph classes provide)."""

    def __init__(self, cbo):
        self.cbo = cbo

    def update_all_do_functions(self, gaussian_processes):
        return [self.update_do_functions(index, gaussian_processes) for index in [0, 1]]

    def update_do_functions(self, index, gaussian_processes):
        return [
            partial(self.update_do_function, gaussian_processes, self.cbo.exploration_set[i], index)
            for i in range(self.cbo.es_size)
        ]

    def update_do_function(self, gaussian_processes, intervention, index, values):
        name = self.cbo.graph.get_gp_name(intervention)
        gp = gaussian_processes[name]
        input_vars = next(filter(lambda dep: dep[0] == intervention[0], self.cbo.graph.fit_dependencies))
        observed = np.hstack([np.asarray(self.cbo.measurements[v], dtype=np.float64).reshape(-1, 1)
                              for v in input_vars])
        intervened_index = [intervention.index(v) if v in intervention else -1 for v in input_vars]
        return do_function(gp, observed, intervened_index, index, values)

    def compute_do(self, measurements, gp, value, input_vars, intervention_vars):
        """DoCalculus.py:68-77 for one value: (mean (N_obs,1), var (N_obs,1)) before the row average."""
        observed = np.hstack([np.asarray(measurements[v], dtype=np.float64).reshape(-1, 1) for v in input_vars])
        idx = [intervention_vars.index(v) if v in intervention_vars else -1 for v in input_vars]
        return gp.predict(intervened_inputs(observed, idx, np.asarray(value, dtype=np.float64)[None, :]))
