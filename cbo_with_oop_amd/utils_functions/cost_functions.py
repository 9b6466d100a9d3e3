"""Cost of an intervention batch on the MI355X path.

Host logic only.  In the reference, ``Cost`` is an emukit ``Acquisition`` whose ``evaluate(x)`` adds up one
cost function per intervened variable, each applied to that variable's whole batch column
(/root/reference/src/utils_functions/cost_functions.py:5-17); with the graph classes' cost functions
(/root/reference/src/graphs/GraphInterface.py:46-50: fixed part plus, for "variable" costs, the sum of
|x| over everything passed in) this yields ONE scalar per batch, and emukit's quotient divides every
Expected-Improvement value of the batch by it.  The HIP sweep takes exactly that scalar
(``cbo_acq_sweep(..., cost, ...)``), so the quirk of summing |x| over the batch is kept.
"""
from __future__ import annotations

import numpy as np


class Cost:
    """Denominator of the acquisition ``EI / Cost`` for one exploration set."""

    has_gradients = True        # the reference reports gradients (all zero) so that L-BFGS may be used

    def __init__(self, costs_functions, evaluated_set):
        self.costs_functions = costs_functions      # mapping variable name -> callable(column) -> scalar
        self.evaluated_set = list(evaluated_set)    # variable names, in the column order of the candidates

    def per_variable(self, x):
        """The individual terms, one per intervened variable (column j of ``x`` belongs to variable j)."""
        x = np.asarray(x)
        return [self.costs_functions[name](x[:, j]) for j, name in enumerate(self.evaluated_set)]

    def evaluate(self, x):
        """Total cost of the batch ``x`` ((M, d)): a scalar (see the module docstring)."""
        return sum(self.per_variable(x), 0)

    def evaluate_with_gradients(self, x):
        """The reference returns zero gradients whatever the cost type (cost_functions.py:23-24)."""
        x = np.asarray(x)
        return self.evaluate(x), np.zeros(x.shape)


def total_cost(intervention_variables, costs, x_new_dict):
    """Cost of performing one intervention, ``x_new_dict`` giving the value set for each intervened variable
    (used by ``CBO.compute_cost``, /root/reference/src/CBO.py:279-291)."""
    return float(sum(costs[name](x_new_dict[name]) for name in intervention_variables))
