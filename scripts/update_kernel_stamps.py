"""Diagnostic build (CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=.../libcbo_hip_diag.so): where the time of trsm_update_kernel goes.
Every workgroup of a launch leaves [start, end] (s_memtime: shader-clock ticks on this chip, one counter per XCD); one
workgroup leaves its stage tops.  Prints the lifetime of a workgroup against its MFMA work (2048 cycles per 16-row stage
and wave: 32 MFMAs x 64) and the probe workgroup's stage-to-stage times -- ~4500 ticks with both workgroups of the CU in
their K-loops, ~2540 while the other slot is empty or in its prologue.  usage: update_kernel_stamps.py [n] [K] [chunk]"""
import ctypes, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cbo_with_oop_amd import _lib
ctx = _lib.Context.get(0); lib = _lib.load()
f = lib.cbo_diag_update_kernel_time
f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_double)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ms = ctypes.c_double()
_lib.check(f(ctx.handle, n, K, chunk, 1, 0, 1, ctypes.byref(ms)))
wg = (ctypes.c_ulonglong * (5 * 65536))(); st = (ctypes.c_ulonglong * 128)()
lib.cbo_diag_upd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.cbo_diag_upd_stamps(wg, st) == 0
w = np.frombuffer(wg, dtype=np.uint64).reshape(65536, 5)
strips, chunks = n // 64, (n - K + 128 * chunk - 1) // (128 * chunk)
nwg = min(strips * chunks, 65536)
w = w[:nwg]
print('unwritten records:', int((w[:, 0] == 0).sum()), 'of', nwg)
w = w[w[:, 0] > 0]
nwg = len(w)
xcc_of = (w[:, 2] >> np.uint64(32)) & np.uint64(0xF)
start = w[:, 0].astype(np.int64).copy(); end = w[:, 1].astype(np.int64).copy()
span = 0
for x in np.unique(xcc_of):                      # s_memtime is per XCD: normalise each XCD to its own first start
    m = xcc_of == x
    b = start[m].min()
    start[m] -= b; end[m] -= b
    span = max(span, int(end[m].max()))
life = end - start
hw = w[:, 2]
cu = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(1 << 16) | (hw & np.uint64(0xFF00)) >> np.uint64(8) | (((hw >> np.uint64(13)) & np.uint64(7)) << np.uint64(8))   # xcc | se | sh/cu bits
life = (end - start).astype(np.float64)
stages = (K // 16) * chunk
print(f"n={n} K={K} chunk={chunk}: {ms.value:.3f} ms per launch (events), {nwg} workgroups of {stages} stages")
print(f"workgroup lifetime (ticks): median {np.median(life):.0f}, p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f}; "
      f"MFMA work of its waves {stages * 2048} cycles -> two resident workgroups use {2 * stages * 2048 / np.median(life):.3f} of the matrix pipe while both are there")
real = (w[:, 4].astype(np.int64) - w[:, 3].astype(np.int64)).astype(np.float64)          # 100 MHz
clock = life.sum() / real.sum() * 100.0
print(f"shader clock over the workgroups' lifetimes: {clock:.0f} MHz (s_memtime ticks per s_memrealtime tick x 100 MHz)")
print(f"slot occupancy: sum of lifetimes / (512 slots x launch time) = {real.sum() / 100.0 / 512 / (ms.value * 1e3):.3f}")
s = np.frombuffer(st, dtype=np.uint64).astype(np.int64)
s = s[s > 0]
if len(s) > 3:
    d = np.diff(s)
    print("probe workgroup, ticks between stamps (C tile loaded, DMA of two stages issued, stage tops ..., last MFMAs issued, stores issued, [next block ...], drained):")
    print(" ".join(f"{x}" for x in d))
