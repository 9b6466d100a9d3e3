"""Where a reference-scale trial's time goes (BASELINE config 1 shape: 2 sets x 200 candidates x 50 observations): the pieces
of the pass timed one by one, 3000 repetitions each, median / mean in microseconds.
usage: python scripts/probes/trial_step_breakdown.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from threadpoolctl import threadpool_limits
from cbo_with_oop_amd import CBOAcquisitionPath, GaussianProcessType, _lib
from cbo_with_oop_amd.graphs import ToyGraph
_limit = threadpool_limits(limits=1)
rng = np.random.default_rng(0)
es = ToyGraph.get_exploration_set("MIS")
xs = [rng.uniform(-5, 5, (50, 1)), rng.uniform(-5, 20, (50, 1))]
ys = [ToyGraph.target_do_x(xs[0]), ToyGraph.target_do_z(xs[1])]
path = CBOAcquisitionPath(GaussianProcessType.NON_CAUSAL_GP, es, ToyGraph.get_cost_structure(1), "min", xs, ys,
                          [ToyGraph.bounds(s) for s in es], grid_shapes=[[200], [200]], comm=None)
path.update_all_gaussian_processes()
best = min(float(ys[0].min()), float(ys[1].min()))
path.last_intervention = 1
path.trial_step(best)
lib = _lib.load()
st = path._call_cache["sweep_sets"]
m1 = path.models[1]
chosen = ctypes.c_int()
yb, bc, vals, idxs = _lib.dptr(st["y_best"]), _lib.dptr(st["batch_cost"]), _lib.dptr(st["vals"]), st["idxs"].ctypes.data_as(_lib.c_int64_p)
Xp, yp = _lib.dptr(m1.X), _lib.dptr(m1._y_flat)
def three_calls():
    path.last_intervention = 1
    path.update_gaussian_process_of_last_intervention()
    _, v = path.compute_best_acquisition_values(best)
    return path.select_next_intervention(v)
def one_call_python():
    path.last_intervention = 1
    return path.trial_step(best)
def one_call_c():
    lib.cbo_trial_step(2, st["gps"], st["cds"], 1, 50, Xp, yp, None, None, yb, 0, 0.0, bc, vals, idxs, ctypes.byref(chosen))
def sweep_only_c():
    lib.cbo_acq_sweep_sets(2, st["gps"], st["cds"], yb, 0, 0.0, bc, vals, idxs)
def upload_only_c():
    lib.cbo_gp_upload_data(m1._handle, 50, Xp, yp, None, None)
def noop_c():
    lib.cbo_abi_version()
for name, fn in (("three calls (round 3's pass)", three_calls), ("path.trial_step", one_call_python), ("cbo_trial_step (C call alone)", one_call_c),
                 ("cbo_acq_sweep_sets alone", sweep_only_c), ("cbo_gp_upload_data alone", upload_only_c), ("a no-op ctypes call", noop_c)):
    for _ in range(200): fn()
    t = np.empty(3000)
    for i in range(3000):
        t0 = time.perf_counter(); fn(); t[i] = time.perf_counter() - t0
    print(f"{name:34s} median {np.median(t) * 1e6:7.1f} us   mean {t.mean() * 1e6:7.1f}   p99 {np.percentile(t, 99) * 1e6:7.1f}")

# ---- the host glue of path.trial_step, statement by statement (the C call excluded)
from cbo_with_oop_amd.utils_functions.utils import winners_to_points
pc = time.perf_counter
acc = np.zeros(8)
reps = 3000
for _ in range(reps):
    path.last_intervention = 1
    t0 = pc()
    s = path.last_intervention
    stc = path._call_cache.get("sweep_sets")
    model = path.models[s]
    fast = (stc is not None and model is not None and path.placement()[0] == "single"
            and model.mean_function is path.mean_functions[s] and model.variance_adjustment is path.var_functions[s]
            and model.hyper_is_initial() and stc["cost_table"] is path.costs and len(stc["models"]) == path.es_size
            and all(a is b for a, b in zip(stc["models"], path.models))
            and all(path._grids.get(i) is not None and path._grids[i][1] is stc["grids"][i] for i in range(path.es_size)))
    t1 = pc()
    model._set_arrays(path.data_x[s], path.data_y[s])
    t2 = pc()
    pm, pv = model._prior(model.X)
    stc["y_best"].fill(best)
    t3 = pc()
    a, b = _lib.dptr(model.X), _lib.dptr(model._y_flat)
    t4 = pc()
    lib.cbo_trial_step(2, stc["gps"], stc["cds"], 1, 50, a, b, None, None, yb, 0, 0.0, bc, vals, idxs, ctypes.byref(chosen))
    t5 = pc()
    xs_, ys_ = winners_to_points(stc, path.models, stc["grids"], best, "min")
    t6 = pc()
    acc[:6] += (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)
print("glue, mean us: conditions %.2f  _set_arrays %.2f  prior+fill %.2f  two pointers %.2f  [C call %.2f]  winners_to_points %.2f"
      % tuple(acc[:6] / reps * 1e6))
