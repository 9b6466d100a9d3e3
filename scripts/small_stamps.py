"""Timing-only (diagnostic build, CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so): s_memtime stamps of workgroup (0, 0) of the
multi-set kernel at BASELINE config 1 shape -- where its time goes, phase by phase (ticks ~ shader clocks)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import CBOAcquisitionPath, GaussianProcessType, _lib
from cbo_with_oop_amd.graphs import ToyGraph
rng = np.random.default_rng(0)
es = ToyGraph.get_exploration_set("MIS")
xs = [rng.uniform(-5, 5, (50, 1)), rng.uniform(-5, 20, (50, 1))]
ys = [ToyGraph.target_do_x(xs[0]), ToyGraph.target_do_z(xs[1])]
path = CBOAcquisitionPath(GaussianProcessType.NON_CAUSAL_GP, es, ToyGraph.get_cost_structure(1), "min", xs, ys,
                          [ToyGraph.bounds(s) for s in es], grid_shapes=[[200], [200]])
path.update_all_gaussian_processes()
best = min(float(ys[0].min()), float(ys[1].min()))
for _ in range(5):
    path.last_intervention = 1
    path.trial_step(best)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
lib.cbo_diag_small_stamps.argtypes = [ctypes.c_void_p]
lib.cbo_diag_small_stamps(buf)
st = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
names = ["descriptor + points -> LDS", "K(X,X) + rhs + zero fill", "factorisation", "factor back to LDS, inverses, z",
         "K(X,X*)", "tile solve", "EI + arg-max", "finish (ticket, reduce, result record)"]
for i, nm in enumerate(names):
    print(f"{nm:40s} {st[i + 1] - st[i]:7d} ticks")
print(f"{'total':40s} {st[8] - st[0]:7d} ticks")
# the factorisation's own stamps (diag128_factor_in_lds: [wave][interval][slot]; every workgroup of the launch writes them,
# whoever came last stays), relative to the start of the factorisation of workgroup (0, 0)
dbuf = (ctypes.c_ulonglong * (4 * 9 * 4))()
lib.cbo_diag_chol_stamps.argtypes = [ctypes.c_void_p]
lib.cbo_diag_chol_stamps(dbuf)
ds = np.frombuffer(dbuf, dtype=np.uint64).astype(np.int64).reshape(4, 9, 4) - st[2]
print("factorisation, cycles since its start; wave 0: [interval] update done / tile factored / stored / -;"
      " waves 1-3: [interval] entered / row panel out / rendezvous passed / trailing done")
for w in range(4):
    print(f"  wave {w}:", "  ".join(str(ds[w, jb].tolist()) for jb in range(4)))
