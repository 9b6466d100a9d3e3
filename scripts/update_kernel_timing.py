"""Diagnostic build (CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so): the LDS-staged update kernel alone at a
given K -- the bulk trailing update of a factorisation: C[K:n, :] -= U[0:K, K:n]^T U[0:K, :], upper part only.
usage: update_kernel_timing.py [n ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbo_with_oop_amd import _lib
ctx = _lib.Context.get(0)
lib = _lib.load()
f = lib.cbo_diag_update_kernel_time
f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_double)]
for n in [int(a) for a in sys.argv[1:]] or [16384, 8192]:
    for upper in (1, 0):
        for half in (1, 0):
            for chunk in (1, 2):
                for klen in (128, 256, 512, 1024):
                    ms = ctypes.c_double()
                    _lib.check(f(ctx.handle, n, klen, chunk, half, upper, 3, ctypes.byref(ms)))
                    rows = n - klen
                    flops = 2.0 * klen * rows * (rows / 2 if upper else n)
                    print(f"n={n} upper={upper} KB={'16' if half else '32'} chunk={chunk} K={klen}: {ms.value:8.3f} ms = {flops / ms.value / 1e9:6.1f} TFLOP/s", flush=True)
