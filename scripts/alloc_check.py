"""Steady-state allocation check: a warm-up, then `iters` rounds of the per-call paths -- L-BFGS
refinement of the grid winner (predict + prediction gradients per iterate), a do-calculus closure call, a predict,
an append + sweep.  Run it twice under `rocprofv3 --hip-trace --stats` with different `iters`: the hipMalloc /
hipFree / hipHostMalloc counts must not depend on `iters`.
usage: python scripts/alloc_check.py [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbo_with_oop_amd import CandidateGrid, CausalExpectedImprovement  # noqa: E402
from cbo_with_oop_amd.DoCalculus import do_function  # noqa: E402
from cbo_with_oop_amd.GaussianProcessFactory import GaussianProcessFactory, GaussianProcessType, HipGaussianProcess  # noqa: E402
from cbo_with_oop_amd.graphs import CompleteGraph  # noqa: E402
from cbo_with_oop_amd.utils_functions import CausalGradientAcquisitionOptimizer, Cost  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(0)
X = rng.uniform([-5, -5], [4, 5], (300, 2))
y = np.sin(X[:, :1]) + 0.1 * X[:, 1:] ** 2 + 0.05 * rng.standard_normal((300, 1))
m = HipGaussianProcess(X[:280], y[:280], noise_var=1e-3)
bounds = CompleteGraph.bounds(["B", "D"])
opt = CausalGradientAcquisitionOptimizer(bounds, grid_shape=[24, 24])
acq = CausalExpectedImprovement(float(y.min()), "min", m) / Cost(CompleteGraph.get_cost_structure(1), ["B", "D"])
obs = rng.normal(size=(200, 3))
gp = GaussianProcessFactory.create(GaussianProcessType.GRAPH_GP, obs, np.sin(obs).sum(1, keepdims=True), [1.0, 1.0, 10.0, False])
vals = rng.uniform(-2, 2, (500, 2))
grid = CandidateGrid(opt.candidates(), m, keep_solution=True)
ei = CausalExpectedImprovement(float(y.min()), "min", m)


def one_round(i):
    opt.optimize(acq, refine=True)
    do_function(gp, obs, [1, -1, 0], 0, vals)
    m.predict(X[:50])
    ei.sweep(grid, cost=2.0)


for i in range(3):
    one_round(i)
for i in range(iters):
    one_round(i)
print("done", iters)
