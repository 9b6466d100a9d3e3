"""BASELINE config 1 shape (toy graph, 50 observations, 200-candidate sweep per set, 2 sets): latency of one
intervene()-style pass (refit one GP, sweep both sets, pick) on the GPU path and on the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CBOAcquisitionPath, GaussianProcessType
from cbo_with_oop_amd.graphs import ToyGraph, meshgrid_candidates
from oracle import gp_oracle as O

rng = np.random.default_rng(0)
es = ToyGraph.get_exploration_set("MIS")
xs = [rng.uniform(-5, 5, (50, 1)), rng.uniform(-5, 20, (50, 1))]
ys = [ToyGraph.target_do_x(xs[0]), ToyGraph.target_do_z(xs[1])]
path = CBOAcquisitionPath(GaussianProcessType.NON_CAUSAL_GP, es, ToyGraph.get_cost_structure(1), "min", xs, ys,
                          [ToyGraph.bounds(s) for s in es], grid_shapes=[[200], [200]])
path.update_all_gaussian_processes()
best = min(float(ys[0].min()), float(ys[1].min()))
path.last_intervention = 1
def gpu_pass():
    path.update_gaussian_process_of_last_intervention()
    xsn, ysn = path.compute_best_acquisition_values(best)
    return path.select_next_intervention(ysn)
grids = [meshgrid_candidates(ToyGraph.bounds(s), [200]) for s in es]
posts = [O.fit(xs[s], ys[s]) for s in range(2)]
def cpu_pass():
    posts[1] = O.fit(xs[1], ys[1])
    vals = [O.acquisition_sweep(posts[s], grids[s], best, cost=1.0)[1] for s in range(2)]
    return O.select_next_intervention([np.array([[v]]) for v in vals])
for name, fn in (("gpu path", gpu_pass), ("cpu oracle", cpu_pass)):
    for _ in range(5): fn()
    t0 = time.perf_counter(); n = 50
    for _ in range(n): r = fn()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.3f} ms per pass -> {400/dt:,.0f} acquisitions/s  (choice {r})")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): gpu_pass()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
