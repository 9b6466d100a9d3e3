"""What one rank of a G-GPU strong-scaling run of a config does, timed on ONE GPU: the config's observations, 1/G of its
fixed candidate grid, the one overlapped call (schedule settled first).  The posterior is replicated, the exchange is one
16-byte all-gather per step (scripts/exchange_latency.py: tens of microseconds), so these times ARE the modelled multi-GPU
step times of DESIGN.md 6; the speed-up column is against the G = 1 row.  usage: python scripts/strong_model.py [c2] [c3]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.sharding import shard_bounds
ctx = _lib.Context.get(0)
lib = _lib.load()
for name in sys.argv[1:] or ["c2", "c3"]:
    cfg = bench.CONFIGS[name]
    X, y, Xs, grid, note = bench.make_problem(cfg, 1, "strong", True)
    y_best = float(y.min())
    model = HipGaussianProcess(X, y, context=ctx, fit=False)
    _lib.check(lib.cbo_gp_fit(model._handle, None, None))
    ctx.set_profiling(True); ctx.reset_timers()
    for _ in range(3):
        _lib.check(lib.cbo_gp_fit(model._handle, None, None))
    fit_ms = ctx.timers()["ms_chol"] / 3
    ctx.set_profiling(False)
    base = None
    print(f"{name}: {X.shape[0]} observations, grid of {Xs.shape[0]} candidates; the factorisation alone {fit_ms:.3f} ms (replicated on every rank)")
    for G in (1, 2, 4, 8):
        b, e = shard_bounds(Xs.shape[0], G, 0)
        cands = CandidateGrid(Xs[b:e], model, index_offset=b, context=ctx)
        bv, bi = ctypes.c_double(), ctypes.c_int64()
        def step():
            _lib.check(lib.cbo_gp_fit_sweep(model._handle, cands._handle, y_best, 0, 0.0, 3.0, None, None, None,
                                            ctypes.byref(bv), ctypes.byref(bi), None, None))
        step()
        calls = 0
        while ctx.schedule_report()[0] > 0 and calls < 90:
            step(); calls += 1
        for _ in range(3):
            step()
        ctx.synchronize(); t0 = time.perf_counter()
        reps = 20 if X.shape[0] <= 4096 else 8
        for _ in range(reps):
            step()
        ctx.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
        base = base or ms
        sched = [l for l in ctx.schedule_report()[1].splitlines() if f"candidates {-(-(e - b) // 64) * 64}:" in l]
        print(f"  G = {G}: {e - b:6d} candidates per rank  {ms:8.3f} ms per step  {Xs.shape[0] / ms / 1e3:8.3f} M acquisitions/s  speed-up {base / ms:5.2f}"
              f"  (Amdahl cap {base / fit_ms:4.1f})  {sched[0].split(';')[1].strip() if sched else ''}", flush=True)
        cands.close()
    model.close()
