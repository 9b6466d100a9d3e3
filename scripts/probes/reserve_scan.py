"""CUs per XCD kept away from the pipelined sweep's streams (CBO_HIP_PIPE_RESERVE, read when the context is made) against
the step time cbo_gp_fit_sweep settles at (the measured schedule adapts the split to each setting).
usage: python scripts/probes/reserve_scan.py [n m]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
lib = _lib.load()
n, mm = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 16384)
bv, bi = ctypes.c_double(), ctypes.c_int64()
rng = np.random.default_rng(0)
X = rng.uniform([-5, -5, -5], [5, 20, 5], (n, 3))
y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
Xs = rng.uniform([-5, -5, -5], [5, 20, 5], (mm, 3))
for reserve in [int(t) for t in os.environ.get("RESERVES", "4,2,3,5,6,8,10,4").split(",")]:
    os.environ["CBO_HIP_PIPE_RESERVE"] = str(reserve)
    ctx = _lib.Context(0)
    del os.environ["CBO_HIP_PIPE_RESERVE"]
    m = HipGaussianProcess(X, y, noise_var=1e-2, fit=False, context=ctx)
    g = CandidateGrid(Xs, m, context=ctx)
    call = lambda: _lib.check(lib.cbo_gp_fit_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, 3.0, None, None, None,
                                                    ctypes.byref(bv), ctypes.byref(bi), None, None))
    call()
    k = 0
    while ctx.schedule_report()[0] > 0 and k < 100:
        call(); k += 1
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            call()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    rep = ctx.schedule_report()[1].split(";")
    print(f"reserve {reserve:2d} CUs/XCD: {best:7.3f} ms/step after {k} settling calls; {rep[1].strip()}", flush=True)
    if os.environ.get("RESERVE_SCAN_FULL"): print("    " + ctx.schedule_report()[1].strip(), flush=True)
    g.close(); m.close(); ctx.close()
