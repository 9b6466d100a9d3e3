"""A consumer of the C-ABI that never calls cbo_shutdown (plain ctypes, not the package's _lib with its Python atexit):
libcbo_hip.so's own atexit handler must release the context's CU-masked streams before the HIP runtime finalises,
otherwise profiling tools that hook finalisation crash (round 1: SIGSEGV in __cxa_finalize under rocprofv3).
Run as:  rocprofv3 --kernel-trace --stats -- python3 scripts/teardown_check.py   (exit code 0, "clean exit" printed)."""
import ctypes
import os

import numpy as np

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cbo_with_oop_amd",
                               "libcbo_hip.so"))
P = ctypes.POINTER(ctypes.c_double)
ctx, gp = ctypes.c_void_p(), ctypes.c_void_p()
assert lib.cbo_init(0, ctypes.byref(ctx)) == 0
rng = np.random.default_rng(0)
X = np.ascontiguousarray(rng.uniform(-3, 3, (1500, 2)))
y = np.ascontiguousarray(np.sin(X).sum(1))
ls = np.array([1.0])
lib.cbo_gp_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, P, P, P, P, ctypes.c_double, P,
                              ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
assert lib.cbo_gp_create(ctx, 0, 1500, 2, X.ctypes.data_as(P), y.ctypes.data_as(P), None, None, 1.0, ls.ctypes.data_as(P), 0,
                         1e-6, 1, ctypes.byref(gp)) == 0
lib.cbo_gp_fit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
assert lib.cbo_gp_fit(gp, None, None) == 0
Xs = np.ascontiguousarray(rng.uniform(-3, 3, (4096, 2)))
bv, bi = ctypes.c_double(), ctypes.c_int64()
lib.cbo_acq_sweep_host.argtypes = [ctypes.c_void_p, ctypes.c_int64, P, P, P, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                   ctypes.c_double, P, P, ctypes.POINTER(ctypes.c_int64)]
assert lib.cbo_acq_sweep_host(gp, 4096, Xs.ctypes.data_as(P), None, None, float(y.min()), 0, 0.0, 2.0, None,
                              ctypes.byref(bv), ctypes.byref(bi)) == 0
print("winner", bi.value, bv.value, "-- exiting WITHOUT cbo_gp_destroy / cbo_shutdown; clean exit")
