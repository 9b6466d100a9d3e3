"""Is the schedule cbo_gp_fit_sweep settles on by measurement (how many panel pairs go right-looking under the
factorisation, in groups or not, the rest in one left-looking launch -- or the plain sequence) the best one?  For every
shape: the step time with the schedule forced (CBO_HIP_PIPE_TAIL = fraction of rows left to the closing launch, x
CBO_HIP_PIPE_GROUP = 1 never grouped / 2 groups of two pairs; CBO_HIP_OVERLAP=0 = the plain sequence) against a context
left to itself until its schedule has settled, all on fresh contexts.
usage: python scripts/schedule_scan.py [NxMxd ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd import CandidateGrid, _lib
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
lib = _lib.load()
shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]] or [
    (2048, 4096, 3), (2048, 16384, 3), (2048, 65536, 3), (2048, 262144, 3), (4096, 4096, 3), (4096, 4096, 1), (4096, 16384, 3),
    (4096, 16384, 1), (4096, 16384, 2), (4096, 65536, 3), (4096, 262144, 3), (4096, 262144, 1), (8192, 4096, 3), (8192, 4096, 2),
    (8192, 16384, 3), (8192, 65536, 3), (8192, 65536, 1), (8192, 262144, 2), (16384, 16384, 3)]
bv, bi = ctypes.c_double(), ctypes.c_int64()


def step_ms(n, mm, d, env):
    """ms per call on a fresh context made under `env`; env None: the context tunes itself first (returns its report too)"""
    for k, v in (env or {}).items():
        os.environ[k] = v
    ctx = _lib.Context(0)
    for k in (env or {}):
        del os.environ[k]
    rng = np.random.default_rng(n + mm + d)
    X = rng.uniform(-5, 5, (n, d)) * (1.0 if d > 1 else 40.0)
    y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    Xs = rng.uniform(-5, 5, (mm, d)) * (1.0 if d > 1 else 40.0)
    m = HipGaussianProcess(X, y, noise_var=1e-2, fit=False, context=ctx)
    g = CandidateGrid(Xs, m, context=ctx)
    call = lambda: _lib.check(lib.cbo_gp_fit_sweep(m._handle, g._handle, float(y.min()), 0, 0.0, float(d), None, None, None,
                                                    ctypes.byref(bv), ctypes.byref(bi), None, None))
    call(); call(); ctx.synchronize()
    report, explored = "", 0
    if env is None:
        while ctx.schedule_report()[0] > 0 and explored < 80:
            call(); explored += 1
        report = ctx.schedule_report()[1].strip()
    reps = 6 if n * mm <= 4096 * 16384 else 3
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    g.close(); m.close(); ctx.close()
    return (best, explored, report) if env is None else best


print("# ms per cbo_gp_fit_sweep call; forced schedules: tail = fraction of the rows left to the closing left-looking launch "
      "(1 = empty pipeline), g1 = updates pair by pair, g2 = in groups of two pairs, seq = the plain sequence", flush=True)
worst = 0.0
for n, mm, d in shapes:
    tails = [1.0, 0.875, 0.75, 0.625, 0.5, 0.375, 0.25, 0.0]
    forced = {"seq": step_ms(n, mm, d, {"CBO_HIP_OVERLAP": "0"})}
    for grp in ("1", "2"):
        for t in tails:
            if t == 1.0 and grp == "2":
                continue
            forced[f"g{grp}/{t}"] = step_ms(n, mm, d, {"CBO_HIP_OVERLAP": "1", "CBO_HIP_PIPE_TAIL": str(t), "CBO_HIP_PIPE_GROUP": grp})
    auto, explored, report = step_ms(n, mm, d, None)
    tb = min(forced, key=forced.get)
    ratio = auto / forced[tb]
    worst = max(worst, ratio)
    print(f"N={n:5d} M={mm:6d} d={d}: auto {auto:8.3f} (settled after {explored} further calls) | best forced {forced[tb]:8.3f} ({tb}) | "
          f"auto/best {ratio:.3f} | " + " ".join(f"{t}:{v:.2f}" for t, v in forced.items()), flush=True)
    print(f"    {report}", flush=True)
print(f"# worst auto/best over the shapes: {worst:.3f}")
