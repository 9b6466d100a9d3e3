"""Monte-Carlo interventional targets on the MI355X.

Mirror of /root/reference/src/utils_functions/graph_functions.py for structural equation models that can be
written as sums of unary functions of earlier nodes plus noise -- the shape of the closed-form SEM the
reference ships (src/graphs/impl/CompleteGraph.py:57-97).  The reference evaluates
``compute_interventions`` as 100 000 Python-level passes over a dict of lambdas (:48-77); here the draws
of all requested interventions are one kernel launch (``cbo_sem_target``).

The noise is NOT generated on the device: the reference seeds numpy's legacy generator
(``np.random.seed(seed)``, :71) and draws ``randn(len(model))`` per sample (:17), which is the row-major
matrix ``RandomState(seed).randn(num_samples, len(model))``.  That matrix is built once on the host with
the same generator, uploaded, and kept resident, so the device averages exactly the reference's draws.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np

from .. import _lib

_HOST_FN = {"id": lambda x: x, "square": lambda x: x * x, "exp": np.exp, "cos": np.cos, "sin": np.sin}


class Term:
    """``c * fn(a * parent)``; fn in {"id", "square", "exp", "cos", "sin"}."""
    __slots__ = ("parent", "fn", "a", "c")

    def __init__(self, parent, fn="id", a=1.0, c=1.0):
        if fn not in _lib.SEM_FN_CODE:
            raise ValueError(f"unknown SEM term function {fn!r}; choose from {sorted(_lib.SEM_FN_CODE)}")
        self.parent, self.fn, self.a, self.c = parent, fn, float(a), float(c)


class AdditiveSEM(OrderedDict):
    """Evaluation-ordered mapping ``node name -> (terms, eps_index)`` (the reference's SEM is an OrderedDict of
    lambdas in the same order, CompleteGraph.py:86-97).  ``eps_index`` is the component of the per-sample noise
    vector the node adds (``epsilon[k]`` in the reference's lambdas), or None."""

    def add(self, name, terms=(), eps=None):
        terms = list(terms)
        for t in terms:
            if t.parent not in self:
                raise ValueError(f"node {name!r} reads {t.parent!r}, which is not an earlier node")
        self[name] = (terms, eps)
        return self

    # ---- device form -----------------------------------------------------------------------------
    def spec(self):
        if len(self) > _lib.SEM_MAX_NODES:
            raise ValueError(f"at most {_lib.SEM_MAX_NODES} nodes")
        order = {name: k for k, name in enumerate(self)}
        sp = _lib.CboSemSpec()
        sp.n_nodes = len(self)
        t = 0
        for k, (name, (terms, eps)) in enumerate(self.items()):
            sp.eps_index[k] = -1 if eps is None else int(eps)
            sp.term_begin[k] = t
            for term in terms:
                if t >= _lib.SEM_MAX_TERMS:
                    raise ValueError(f"at most {_lib.SEM_MAX_TERMS} terms")
                sp.term_parent[t] = order[term.parent]
                sp.term_fn[t] = _lib.SEM_FN_CODE[term.fn]
                sp.term_a[t] = term.a
                sp.term_c[t] = term.c
                t += 1
        for k in range(len(self), _lib.SEM_MAX_NODES + 1):
            sp.term_begin[k] = t
        return sp

    def device(self, num_samples=100000, seed=1, context=None):
        """The resident (noise matrix, model) pair for one (num_samples, seed); cached on the model.  A model that
        carries its own draws (``from_linear``) ignores ``num_samples`` / ``seed``: the draws ARE the samples."""
        cache = self.__dict__.setdefault("_device", {})
        draws = self.__dict__.get("draws")
        key = ("own", id(context)) if draws is not None else (int(num_samples), int(seed), id(context))
        if key not in cache:
            cache[key] = DeviceSEM(self, num_samples, seed, context, draws=draws)
        return cache[key]

    @classmethod
    def from_linear(cls, order, parents, coefs, intercepts, exogenous):
        """A SEM whose endogenous nodes are linear regressions of earlier nodes and whose exogenous nodes are draws the
        caller made -- the shape of /root/reference/src/graphs/impl/CoralGraph.py:91-160 (``LinearRegression`` per node,
        :91-94, evaluated as ``regressions[v].predict(parents)`` in ``define_sem``, :104-160; ``N`` and ``L`` drawn from
        a Gaussian mixture / a gamma fitted to the data, :96-101, :106-110).

        order       node names in evaluation order
        parents     {node: [parent names]} for the regression nodes (``var_dependencies``, CoralGraph.py:40-50)
        coefs       {node: coefficients in the order of parents[node]} (``LinearRegression.coef_``)
        intercepts  {node: float} (``LinearRegression.intercept_``)
        exogenous   {node: 1-D array of num_samples draws} for the nodes without parents; the reference draws one
                    value per sample from scipy / sklearn's global generators, so the caller makes them

        A regression node is ``sum_i coef_i * parent_i + intercept``, terms left to right, the intercept last
        (``X @ coef_ + intercept_``).  The intercept rides in the node's noise slot: the draws matrix the model carries
        has one column per exogenous node (its draws) and one constant column per regression node."""
        model = cls()
        exo = [n for n in order if n not in parents]
        missing = [n for n in exo if n not in exogenous]
        if missing:
            raise ValueError(f"no draws for the exogenous node(s) {missing}")
        num = {len(np.asarray(exogenous[n]).reshape(-1)) for n in exo}
        if len(num) != 1:
            raise ValueError("every exogenous node needs the same number of draws")
        num_samples = num.pop()
        cols = []
        for name in order:
            if name in parents:
                c = np.asarray(coefs[name], dtype=np.float64).reshape(-1)
                if len(c) != len(parents[name]):
                    raise ValueError(f"node {name!r}: {len(c)} coefficients for {len(parents[name])} parents")
                model.add(name, [Term(p, "id", 1.0, float(ci)) for p, ci in zip(parents[name], c)], eps=len(cols))
                cols.append(np.full(num_samples, float(np.asarray(intercepts[name]).reshape(-1)[0])))
            else:
                model.add(name, [], eps=len(cols))
                cols.append(np.asarray(exogenous[name], dtype=np.float64).reshape(-1))
        model.__dict__["draws"] = np.ascontiguousarray(np.stack(cols, axis=1))
        return model


def reference_noise(num_samples, n_nodes, seed):
    """The draws of compute_interventions (graph_functions.py:71, :17), one row per sample."""
    return np.random.RandomState(seed).randn(int(num_samples), int(n_nodes))


class DeviceSEM:
    """cbo_sem handle: the model and its noise matrix on the device."""

    def __init__(self, model, num_samples=100000, seed=1, context=None, draws=None):
        self.model = model
        self.names = list(model)
        self.ctx = context or _lib.Context.get()
        self._lib = _lib.load()
        if draws is None:
            eps = np.ascontiguousarray(reference_noise(num_samples, len(model), seed))
        else:                                              # the caller's own samples, one row per draw
            eps = np.ascontiguousarray(draws, dtype=np.float64)
            num_samples = eps.shape[0]
        self.num_samples = int(num_samples)
        self._spec = model.spec()
        h = ctypes.c_void_p()
        _lib.check(self._lib.cbo_sem_create(self.ctx.handle, ctypes.byref(self._spec), self.num_samples,
                                            eps.shape[1], _lib.dptr(eps), ctypes.byref(h)))
        self._handle = h

    def target_means(self, intervened, values, target_variable="Y"):
        """Mean of ``target_variable`` under do(intervened = values[i]) for every row i of ``values`` ((M, len(intervened)))."""
        nodes = np.asarray([self.names.index(n) for n in intervened], dtype=np.int32)
        values = _lib.as_f64(values).reshape(-1, max(1, len(nodes))) if len(nodes) else np.zeros((1, 1))
        m = values.shape[0]
        out = np.empty(m)
        _lib.check(self._lib.cbo_sem_target(self._handle, self.names.index(target_variable), m, len(nodes),
                                            nodes.ctypes.data_as(_lib.c_int_p) if len(nodes) else None,
                                            _lib.dptr(values) if len(nodes) else None, _lib.dptr(out)))
        return out.reshape(m, 1)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h and not self.ctx.closed:
            try:
                self._lib.cbo_sem_destroy(h)
            except Exception:
                pass


class _Intervened(AdditiveSEM):
    """Result of intervene_dict: the model plus the values its intervened nodes are clamped to."""
    fixed = None


def intervene_dict(model, **interventions):
    """graph_functions.py:30-45: a copy of the model whose intervened nodes return their intervention value."""
    new_model = _Intervened(model)
    new_model.__dict__.update(model.__dict__)
    new_model.fixed = dict(getattr(model, "fixed", None) or {}, **interventions)
    return new_model


def sample_from_model(model, epsilon=None):
    """graph_functions.py:8-27: one draw, on the host (the device path is compute_interventions)."""
    epsilon = np.random.randn(len(model)) if epsilon is None else epsilon
    fixed = getattr(model, "fixed", None) or {}
    sample = OrderedDict()
    for name, (terms, eps) in model.items():
        if name in fixed:
            sample[name] = fixed[name]
            continue
        v = None
        for t in terms:
            g = t.c * _HOST_FN[t.fn](t.a * sample[t.parent])
            v = g if v is None else v + g
        if eps is not None:
            v = epsilon[eps] if v is None else v + epsilon[eps]
        sample[name] = v
    return sample


def compute_interventions(model, interventions, node_values, target_variable="Y", num_samples=100000, seed=1):
    """graph_functions.py:48-77 with the same signature, side effect (``interventions`` receives the values) and
    return shape (1, 1).  ``node_values`` may hold several rows: row i is one intervention and the result is
    (M, 1) -- the batched form the reference loops over."""
    node_values = np.atleast_2d(np.asarray(node_values, dtype=np.float64))
    names = list(interventions.keys())
    for i, node in enumerate(names):
        interventions[node] = node_values[0, i]
    values = node_values[:, :len(names)]
    clamped = getattr(model, "fixed", None) or {}          # a model that went through intervene_dict already
    if clamped:
        extra = [n for n in clamped if n not in names]
        values = np.hstack([np.tile([float(clamped[n]) for n in extra], (values.shape[0], 1)), values])
        names = extra + names
    return model.device(num_samples, seed).target_means(names, values, target_variable)


def get_parameter_space(interventions, min_interventions, max_interventions):
    """graph_functions.py:80-93 without emukit: the [(lo, hi)] list every consumer of the space here accepts
    (``find_next_y_point``, ``space_bounds``)."""
    return [(float(lo), float(hi)) for _, lo, hi in zip(interventions.keys(), min_interventions, max_interventions)]
