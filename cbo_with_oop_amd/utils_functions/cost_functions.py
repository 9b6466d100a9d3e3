"""Cost of an intervention batch, mirroring /root/reference/src/utils_functions/cost_functions.py:5-31.

Host logic only: ``Cost.evaluate`` reduces a whole batch to ONE scalar (the reference's variable costs
sum |x| over the batch column, cost_functions.py:16 + GraphInterface.py:48-49), and that scalar is what
the HIP sweep divides the Expected Improvement by.
"""
from __future__ import annotations

import numpy as np


class Cost:
    def __init__(self, costs_functions, evaluated_set):
        self.costs_functions = costs_functions
        self.evaluated_set = evaluated_set

    def evaluate(self, x):
        cost = 0
        for i in range(len(self.evaluated_set)):
            cost += self.costs_functions[self.evaluated_set[i]](x[:, i])
        return cost

    @property
    def has_gradients(self):
        return True

    def evaluate_with_gradients(self, x):
        return self.evaluate(x), np.zeros(x.shape)


def total_cost(intervention_variables, costs, x_new_dict):
    """cost_functions.py:27-31."""
    cost = 0.
    for i in range(len(intervention_variables)):
        cost += costs[intervention_variables[i]](x_new_dict[intervention_variables[i]])
    return cost
