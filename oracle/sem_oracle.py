"""CPU restatement of the reference's Monte-Carlo interventional target -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (cbo_with_oop_amd/) never does.

PARITY UNPINNED against a run of the reference: its graph classes import GPy/emukit/sklearn, none of
which is installed here.  What pins it instead: (1) the closed-form expectations of the complete graph's
target under interventions that cut every noisy ancestor (tests/test_oracle.py), (2) the reference's own
data/complete_graph/interventional_data_{x,y}_BO.npy (columns B, E, D -> mean Y), which these means
reproduce to Monte-Carlo accuracy (tests/golden/complete_bo_d3.npz), (3) toy_graph's shipped noise-free
curves (4e-14).

Follows /root/reference/src/utils_functions/graph_functions.py:
  sample_from_model      :8-27   one pass over the OrderedDict of node functions
  intervene_dict         :30-45  intervened nodes become constants
  compute_interventions  :48-77  seed, num_samples draws, mean of the target column
and the structural equations of /root/reference/src/graphs/impl/CompleteGraph.py:57-97.
"""
from collections import OrderedDict

import numpy as np


def complete_graph_sem():
    """CompleteGraph.define_sem (CompleteGraph.py:57-97): node -> f(epsilon, values so far)."""
    return OrderedDict([
        ("U1", lambda e, v: e[0]),
        ("U2", lambda e, v: e[1]),
        ("F", lambda e, v: e[8]),
        ("A", lambda e, v: v["F"] ** 2 + v["U1"] + e[2]),
        ("B", lambda e, v: v["U2"] + e[3]),
        ("C", lambda e, v: np.exp(-v["B"]) + e[4]),
        ("D", lambda e, v: np.exp(-v["C"]) / 10. + e[5]),
        ("E", lambda e, v: np.cos(v["A"]) + v["C"] / 10. + e[6]),
        ("Y", lambda e, v: np.cos(v["D"]) - v["D"] / 5. + np.sin(v["E"]) - v["E"] / 4. + v["U1"]
            + np.exp(-v["U2"]) + e[7]),
    ])


def toy_graph_sem():
    """Noisy form of the toy relations (the reference ships the data but no class)."""
    return OrderedDict([
        ("X", lambda e, v: e[0]),
        ("Z", lambda e, v: np.exp(-v["X"]) + e[1]),
        ("Y", lambda e, v: np.cos(v["Z"]) - np.exp(-v["Z"] / 20.) + e[2]),
    ])


def linear_sem(order, parents, coefs, intercepts):
    """CoralGraph.define_sem (/root/reference/src/graphs/impl/CoralGraph.py:104-160) for given regressions: a node with
    parents is ``regressions[v].predict(hstack(parents))`` = parents @ coef_ + intercept_ (:112-146); a node without
    is the caller's draw for that sample (``dist.sample(1)`` / ``dist.rvs(1)`` there, :106-110), taken from column k of
    the draws matrix, k = the node's position among the exogenous nodes."""
    sem, k = OrderedDict(), 0
    for name in order:
        if name in parents:
            ps, c, b = list(parents[name]), np.asarray(coefs[name], dtype=np.float64).reshape(-1), float(intercepts[name])

            def f(e, v, ps=ps, c=c, b=b):
                acc = c[0] * v[ps[0]]
                for ci, p in zip(c[1:], ps[1:]):
                    acc = acc + ci * v[p]
                return acc + b
            sem[name] = f
        else:
            sem[name] = (lambda e, v, k=k: e[k])
            k += 1
    return sem


def sample_from_model(sem, fixed, epsilon):
    """One draw (graph_functions.py:8-27 on the mutilated model of :30-45)."""
    values = OrderedDict()
    for name, f in sem.items():
        values[name] = fixed[name] if name in fixed else f(epsilon, values)
    return values


def compute_interventions_loop(sem, fixed, target="Y", num_samples=100000, seed=1):
    """graph_functions.py:48-77 literally: seed the legacy generator, draw len(model) normals per sample."""
    np.random.seed(seed)
    ys = [sample_from_model(sem, fixed, np.random.randn(len(sem)))[target] for _ in range(num_samples)]
    return float(np.mean(np.asarray(ys, dtype=np.float64)))


def compute_interventions(sem, fixed, target="Y", num_samples=100000, seed=1, draws=None):
    """Same numbers with the draws stacked: the node functions are elementwise, so passing the noise matrix's
    columns evaluates every draw at once (equal to the loop draw for draw; the mean's summation order is
    numpy's pairwise one in both).  ``draws``: the caller's own sample matrix instead of the seeded normals."""
    eps = np.random.RandomState(seed).randn(num_samples, len(sem)) if draws is None else np.asarray(draws, dtype=np.float64)
    num_samples = eps.shape[0]
    cols = [eps[:, k] for k in range(eps.shape[1])]
    values = OrderedDict()
    for name, f in sem.items():
        values[name] = np.full(num_samples, float(fixed[name])) if name in fixed else f(cols, values)
    return float(np.mean(values[target]))
