"""A CBO run on the reference's complete graph through the mirrored stack (per-set GPs, grid acquisition per set, set
selection, Monte-Carlo target of the chosen intervention with the reference's 100 000 draws, data append), timed per
trial on the device and on the CPU restatements (the reference itself needs GPy/emukit).  Same choices on both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CBOAcquisitionPath, GaussianProcessType
from cbo_with_oop_amd.graphs import CompleteGraph, meshgrid_candidates
from cbo_with_oop_amd.utils_functions import graph_functions as G
from oracle import gp_oracle as O, sem_oracle as S

TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
# --optimize: the reference's loop as it is written -- after the new observation is added, the intervened set's model
# runs its hyper-parameter MLE (src/CBO.py:173), and the NEXT trial rebuilds that model with the constructor's
# hyper-parameters (src/CBO.py:224-235), i.e. the optimised values are used for nothing but are paid for every trial
OPTIMIZE = "--optimize" in sys.argv
es = CompleteGraph.get_exploration_set("MIS")
bounds = [CompleteGraph.bounds(s) for s in es]
shapes = [[200] if len(s) == 1 else [32, 32] for s in es]
costs = CompleteGraph.get_cost_structure(1)
sem_dev, sem_cpu = CompleteGraph.define_sem(), S.complete_graph_sem()
rng = np.random.default_rng(0)
x0 = [np.array([[rng.uniform(lo, hi) for lo, hi in b] for _ in range(5)]) for b in bounds]
y0 = [G.compute_interventions(sem_dev, {n: "" for n in es[s]}, x0[s]) for s in range(len(es))]

def run(device):
    xs, ys = [x.copy() for x in x0], [y.copy() for y in y0]
    best = min(float(y.min()) for y in ys)
    grids = [meshgrid_candidates(bounds[s], shapes[s]) for s in range(len(es))]
    if device:
        path = CBOAcquisitionPath(GaussianProcessType.NON_CAUSAL_GP, es, costs, "min", xs, ys, bounds, grid_shapes=shapes)
        path.update_all_gaussian_processes()
    trace, times = [], []
    for _ in range(TRIALS):
        t0 = time.perf_counter()
        if device:
            x_new, y_acq = path.compute_best_acquisition_values(best)
            _, s = path.select_next_intervention(y_acq)
            x_pick = x_new[s]
            y_new = G.compute_interventions(sem_dev, {n: "" for n in es[s]}, x_pick)
        else:
            vals, idxs = [], []
            for k in range(len(es)):
                _, val, idx, _, _ = O.acquisition_sweep(O.fit(xs[k], ys[k]), grids[k], best, cost=float(len(es[k])))
                vals.append(val); idxs.append(idx)
            s = O.select_next_intervention([np.array([[v]]) for v in vals])
            x_pick = grids[s][idxs[s]][None, :]
            y_new = np.array([[S.compute_interventions(sem_cpu, dict(zip(es[s], x_pick[0])))]])
        xs[s] = np.vstack([xs[s], x_pick]); ys[s] = np.vstack([ys[s], y_new])
        best = min(best, float(y_new[0, 0]))
        if device:
            path.data_x[s], path.data_y[s] = xs[s], ys[s]
            if OPTIMIZE:
                path.models[s].set_data(xs[s], ys[s])            # Monitor.add_intervention_data -> model.set_data
                path.models[s].optimize()                        # CBO.py:173
            path.update_gaussian_process_of_last_intervention()  # next trial's rebuild (constructor hyper-parameters)
        elif OPTIMIZE:
            O.optimize_hyperparameters(xs[s], ys[s])
        times.append(time.perf_counter() - t0)
        trace.append((s, tuple(np.round(x_pick[0], 10))))
    return trace, times, best

for name, dev in (("device", True), ("cpu restatements", False)):
    trace, times, best = run(dev)
    print(f"{name}{' (with the per-trial optimize())' if OPTIMIZE else ''}: median {np.median(times[2:])*1e3:.2f} ms per trial over {TRIALS} trials, best target {best:.4f}, "
          f"sets chosen {[s for s, _ in trace]}", flush=True)
    if dev:
        dev_trace = trace
print("same choices:", dev_trace == trace)
