"""Problem tables the hot path needs from the reference's graph classes: exploration sets,
interventional ranges and cost constants, plus the closed-form structural equation models in the
additive form the device Monte-Carlo target takes.  The graph-level GP fits and the
data-fitted SEMs of the coral graphs (sklearn regressors and mixtures fitted to SEM_data.mat,
src/graphs/impl/CoralGraph.py:103-159) are out of scope.

Citations are relative to /root/reference/.
"""
from __future__ import annotations

from collections import OrderedDict
from functools import partial

import numpy as np


def cost(fix_cost, variable_cost, intervention_value, **kwargs):
    """GraphInterface.cost (src/graphs/GraphInterface.py:46-50).  The variable part sums |x| over
    everything it is given -- for a batch column that is a scalar for the whole batch."""
    total_cost = fix_cost
    if variable_cost is True:
        total_cost += np.sum(np.abs(intervention_value))
    return total_cost


def _cost_table(names, fixed, variable):
    return OrderedDict((n, partial(cost, f, variable)) for n, f in zip(names, fixed))


class _Graph:
    name = ""
    variables = ()
    _fix_different = ()
    _ranges = OrderedDict()
    _mis = ()
    _pomis = ()

    @classmethod
    def get_exploration_set(cls, set_name="MIS"):
        return [list(s) for s in (cls._mis if set_name == "MIS" else cls._pomis)]

    @classmethod
    def get_interventional_ranges(cls):
        return OrderedDict((k, list(v)) for k, v in cls._ranges.items())

    @classmethod
    def get_cost_structure(cls, type_cost):
        """Same four cost types as CompleteGraph.py:139-181 / CoralGraph.py:211-252."""
        ones = [1] * len(cls.variables)
        if type_cost == 1:
            return _cost_table(cls.variables, ones, False)
        if type_cost == 2:
            return _cost_table(cls.variables, cls._fix_different, False)
        if type_cost == 3:
            return _cost_table(cls.variables, cls._fix_different, True)
        if type_cost == 4:
            return _cost_table(cls.variables, ones, True)
        raise RuntimeError(f"[ERROR] Invalid cost type: {type_cost}")

    @classmethod
    def bounds(cls, variables):
        """[(lo, hi)] for an exploration set -- what graph_functions.get_parameter_space
        (src/utils_functions/graph_functions.py:80-93) turns into an emukit ParameterSpace."""
        return [tuple(cls._ranges[v]) for v in variables]


class CompleteGraph(_Graph):
    """src/graphs/impl/CompleteGraph.py:100-112, 139-181."""
    name = "complete_graph"
    variables = ("A", "B", "C", "D", "E", "F")
    _fix_different = (1, 10, 2, 5, 20, 3)
    _ranges = OrderedDict([("E", (-6, 3)), ("B", (-5, 4)), ("D", (-5, 5)), ("F", (-4, 4))])
    _mis = (("B",), ("D",), ("E",), ("B", "D"), ("B", "E"), ("D", "E"))
    _pomis = (("B",), ("D",), ("E",), ("B", "D"), ("D", "E"))

    @staticmethod
    def define_sem():
        """src/graphs/impl/CompleteGraph.py:57-97 as an AdditiveSEM (same node order, same noise components).
        ``/10.`` and ``/5.`` are written as multiplications by 0.1 and 0.2: at most one ulp per draw away from
        the reference's divisions, far below the Monte-Carlo error of the mean."""
        from .utils_functions.graph_functions import AdditiveSEM, Term as T
        sem = AdditiveSEM()
        sem.add("U1", eps=0).add("U2", eps=1).add("F", eps=8)
        sem.add("A", [T("F", "square"), T("U1")], eps=2)
        sem.add("B", [T("U2")], eps=3)
        sem.add("C", [T("B", "exp", a=-1.0)], eps=4)
        sem.add("D", [T("C", "exp", a=-1.0, c=0.1)], eps=5)
        sem.add("E", [T("A", "cos"), T("C", c=0.1)], eps=6)
        sem.add("Y", [T("D", "cos"), T("D", c=-0.2), T("E", "sin"), T("E", c=-0.25), T("U1"),
                      T("U2", "exp", a=-1.0)], eps=7)
        return sem


class CoralGraph(_Graph):
    """src/graphs/impl/CoralGraph.py:162-184, 211-252."""
    name = "coral_graph"
    variables = ("N", "O", "C", "T", "D")
    _fix_different = (1, 10, 2, 5, 20)
    _ranges = OrderedDict([("N", (-2, 5)), ("O", (2, 4)), ("C", (0, 1)), ("T", (2450, 2500)),
                           ("D", (1950, 1965))])
    _mis = (
        ("N",), ("O",), ("C",), ("T",), ("D",),
        ("N", "O"), ("N", "C"), ("N", "T"), ("N", "D"), ("O", "C"), ("O", "T"), ("O", "D"),
        ("T", "C"), ("T", "D"), ("C", "D"),
        ("N", "O", "C"), ("N", "O", "T"), ("N", "O", "D"), ("N", "C", "T"), ("N", "C", "D"),
        ("N", "T", "D"), ("O", "C", "T"), ("O", "C", "D"), ("C", "T", "D"), ("O", "T", "D"),
    )
    _pomis = _mis
    # evaluation order of define_sem (CoralGraph.py:148-160) and the regressions' inputs (var_dependencies, :40-50)
    sem_order = ("N", "L", "TE", "C", "S", "T", "D", "P", "O", "CO", "Y")
    var_dependencies = OrderedDict([
        ("Y", ["L", "N", "P", "O", "C", "CO", "TE"]), ("P", ["S", "T", "D", "TE"]), ("O", ["S", "T", "D", "TE"]),
        ("CO", ["S", "T", "D", "TE"]), ("T", ["S"]), ("D", ["S"]), ("C", ["N", "L", "TE"]), ("S", ["TE"]), ("TE", ["L"]),
    ])

    @classmethod
    def define_sem(cls, coefs, intercepts, exogenous):
        """CoralGraph.define_sem (CoralGraph.py:104-160) from what a reference process holds after its constructor:
        ``coefs[v]`` / ``intercepts[v]`` = ``self.regressions[v].coef_`` / ``.intercept_`` (:91-94, fitted there on
        ``true_observations.pkl``, a pickled DataFrame this package does not load) and ``exogenous`` = {"N": draws of
        ``dist_nutrients_pc1.sample``, "L": draws of ``dist_Light.rvs``} (:96-101), one per Monte-Carlo sample."""
        from .utils_functions.graph_functions import AdditiveSEM
        return AdditiveSEM.from_linear(cls.sem_order, cls.var_dependencies, coefs, intercepts, exogenous)


class SimplifiedCoralGraph(CoralGraph):
    """src/graphs/impl/SimplifiedCoralGraph.py:185-192: CoralGraph's variables and exploration sets, its own (narrower)
    interventional ranges."""
    name = "simplified_coral_graph"
    _ranges = OrderedDict([("N", (-2, 5)), ("O", (3, 4)), ("C", (0.3, 0.4)), ("T", (2300, 2400)),
                           ("D", (2000, 2080))])


class ToyGraph(_Graph):
    """The reference ships ``data/toy_graph`` but no graph class, so
    ``runCBO.py --experiment toy_graph`` cannot run there.  This counterpart is derived from the
    shipped data: X -> Z -> Y with Z = exp(-X), Y = cos(Z) - exp(-Z/20) (noise-free means; holds to
    4e-14 on data/toy_graph/interventional_data_{x,y}_BO.npy).  Ranges X in [-5,5], Z in [-5,20] are
    the upstream CBO values (the BO grid spans Z in [-5,20]); a third manipulative dimension does
    not exist in toy_graph, so BASELINE.json's d=3 "toy_graph" config uses the box
    [-5,5] x [-5,20] x [-5,5] with a smooth synthetic target (bench.py documents it)."""
    name = "toy_graph"
    variables = ("X", "Z")
    _fix_different = (1, 1)
    _ranges = OrderedDict([("X", (-5, 5)), ("Z", (-5, 20))])
    _mis = (("X",), ("Z",))
    _pomis = (("Z",),)

    @staticmethod
    def define_sem():
        """X = e0, Z = exp(-X) + e1, Y = cos(Z) - exp(-Z/20) + e2: the noisy form of the relations above
        (upstream CBO's toy graph; not in the reference, see the class docstring)."""
        from .utils_functions.graph_functions import AdditiveSEM, Term as T
        sem = AdditiveSEM()
        sem.add("X", eps=0)
        sem.add("Z", [T("X", "exp", a=-1.0)], eps=1)
        sem.add("Y", [T("Z", "cos"), T("Z", "exp", a=-0.05, c=-1.0)], eps=2)
        return sem

    @staticmethod
    def target_do_z(z):
        z = np.asarray(z, dtype=np.float64)
        return np.cos(z) - np.exp(-z / 20.0)

    @staticmethod
    def target_do_x(x):
        return ToyGraph.target_do_z(np.exp(-np.asarray(x, dtype=np.float64)))


GRAPHS = {g.name: g for g in (CompleteGraph, CoralGraph, SimplifiedCoralGraph, ToyGraph)}


def meshgrid_candidates(bounds, points_per_dim):
    """Regular grid over a box, first dimension slowest -- how the reference's stored interventional
    grids are laid out (linspace per variable, cartesian product)."""
    axes = [np.linspace(lo, hi, n) for (lo, hi), n in zip(bounds, points_per_dim)]
    mesh = np.meshgrid(*axes, indexing="ij")
    return np.ascontiguousarray(np.stack([m.reshape(-1) for m in mesh], axis=1))
