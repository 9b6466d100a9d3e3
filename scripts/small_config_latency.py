"""BASELINE config 1 shape (toy graph, 50 observations, 200-candidate sweep per set, 2 sets): latency of one
intervene()-style pass (refit one GP, sweep both sets, pick) on the GPU path and on the CPU oracle.

The 50-70 ms outliers of rounds 1-2 (one call in ~1000) were this script's own doing: the CPU-oracle passes leave numpy's
BLAS pool (one thread per visible core: 256 on the GPU box) spinning, the box's cgroup allows 16 CPUs per 100 ms period
(/sys/fs/cgroup/cpu.max = 1600000 100000), and the kernel's bandwidth control then freezes EVERY thread of the process
until the period ends -- including the one timing the library (CBO_HIP_TRACE_SLOW=1 showed the stall inside a plain
spin on pinned memory, not in any runtime call).  The device timings below therefore run with the BLAS pool limited to one
thread (threadpoolctl); `--oversubscribe` keeps the pool as it is and shows the stalls again, with the cgroup's
nr_throttled count before and after."""
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
from cbo_with_oop_amd.utils_functions.utils import find_next_y_point
def gpu_pass_per_set():
    """the same pass with one device call (and one synchronisation) per set: round 1's path"""
    path.update_gaussian_process_of_last_intervention()
    ysn = [find_next_y_point(path.space_list[s], path.models[s], best, es[s], path.costs, task="min",
                             candidates=path.candidate_grid(s))[0] for s in range(2)]
    return path.select_next_intervention(ysn)
def device_only():
    """just the multi-set device call (no upload of the intervened set, no host bookkeeping)"""
    return path.compute_best_acquisition_values(best)[1][0][0, 0]
def throttle_count():
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("nr_throttled"):
                return int(line.split()[1])
    except OSError:
        pass
    return None
from threadpoolctl import threadpool_limits
_limit = None if "--oversubscribe" in sys.argv else threadpool_limits(limits=1)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), "| visible cores:", os.cpu_count(),
          "| BLAS pool:", "unlimited (--oversubscribe)" if _limit is None else "1 thread")
except OSError:
    pass
thr0 = throttle_count()
for name, fn in (("gpu path (one launch for all sets)", gpu_pass), ("gpu path, per-set calls", gpu_pass_per_set),
                 ("  of which cbo_acq_sweep_sets + host glue", device_only), ("cpu oracle", cpu_pass)):
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
# the C-ABI call alone (no Python glue): cbo_acq_sweep_sets on the two sets
import ctypes
from cbo_with_oop_amd import _lib
lib = _lib.load()
grids_dev = [path.candidate_grid(s) for s in range(2)]
gps = (ctypes.c_void_p * 2)(*[m._handle for m in path.models])
cds = (ctypes.c_void_p * 2)(*[g._handle for g in grids_dev])
yb, cs, vals, idxs = np.full(2, best), np.ones(2), np.empty(2), np.empty(2, dtype=np.int64)
args = (2, gps, cds, _lib.dptr(yb), 0, 0.0, _lib.dptr(cs), _lib.dptr(vals), idxs.ctypes.data_as(_lib.c_int64_p))
for _ in range(5): _lib.check(lib.cbo_acq_sweep_sets(*args))
ts = []
for _ in range(2000):
    t0 = time.perf_counter(); lib.cbo_acq_sweep_sets(*args); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print(f"  cbo_acq_sweep_sets alone (2 sets x 200 candidates, 50 observations each): median {np.median(ts):.1f} us, mean "
      f"{ts.mean():.1f} us, max {ts.max():.0f} us over {len(ts)} calls ({int((ts > 1000).sum())} above 1 ms)")
thr1 = throttle_count()
if thr0 is not None:
    print(f"  cgroup periods in which this process group was throttled while the script ran: {thr1 - thr0}")
