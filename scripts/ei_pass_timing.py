"""The EI / cost / arg-max pass alone (acq_kernel + argmax_final_kernel on stored q, mu), as bench.py's roofline_ei times it:
ms per pass and fraction of the 8 TB/s HBM roofline at 3 doubles per candidate.  usage: python scripts/ei_pass_timing.py [log2m ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
ctx = _lib.Context.get(0)
lib = _lib.load()
Xe = np.random.default_rng(3).uniform(-5.0, 5.0, (64, 3))
ye = np.sin(Xe).sum(1, keepdims=True)
me = HipGaussianProcess(Xe, ye, context=ctx)
for log2m in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    m = 1 << log2m
    Ce = np.random.default_rng(4).uniform(-5.0, 5.0, (m, 3))
    ge = CandidateGrid(Ce, me, context=ctx)
    acq = np.empty(m)
    bv, bi = ctypes.c_double(), ctypes.c_int64()
    def ei_pass(out=True):
        _lib.check(lib.cbo_acq_sweep(me._handle, ge._handle, float(ye.min()), 0, 0.0, 3.0, _lib.dptr(acq) if out else None, None, None,
                                     ctypes.byref(bv), ctypes.byref(bi)))
    for with_out in (True, False):
        ei_pass(with_out); ei_pass(with_out)
        ctx.set_profiling(True); ctx.reset_timers()
        for _ in range(10):
            ei_pass(with_out)
        t = ctx.timers(); ctx.set_profiling(False)
        ms = t["ms_acq"] / 10
        nbytes = (24.0 if with_out else 16.0) * m
        print(f"2^{log2m} candidates, acquisition values {'stored' if with_out else 'not stored'}: {ms * 1e3:8.1f} us per pass = {nbytes / ms / 1e6:7.1f} GB/s"
              f" = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s; winner {bi.value} {bv.value:.6e}", flush=True)
    ge.close()
