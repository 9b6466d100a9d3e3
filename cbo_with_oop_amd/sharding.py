"""Candidate-grid sharding across the GPUs of one node.

The acquisition sweep is independent per candidate once the posterior is fitted, so the grid is cut
into contiguous blocks, one per rank (one process per GPU); every rank fits the same posterior from
the same (X, y) (replicated, bit-identical, no data-path collective) and sweeps its block.  The only
exchange is the arg-max: each rank contributes (best_val, best_global_idx) = 16 bytes, all-gathered
through the launcher's process group -- with backend "nccl" that is RCCL over xGMI -- and every rank
reduces the gathered pairs with the same tie rule (lowest global index wins; NaN maximal), so all
ranks agree on the winner.  RCCL has no MAXLOC, hence gather + local reduce.

torch.distributed is used for the rendezvous/collective only (plumbing); no torch types cross the
C-ABI.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def shard_bounds(m, world_size, rank):
    """Contiguous block [begin, end) of an m-row grid for ``rank``: ceil(m / world) rows each, the
    last ranks may get fewer (or none)."""
    per = -(-m // world_size)
    begin = min(m, rank * per)
    return begin, min(m, begin + per)


def reduce_pairs(vals, idxs):
    """Global winner of gathered (value, global index) pairs via the C-ABI's host reduction
    (``cbo_argmax_pairs``): same comparator as the device arg-max."""
    vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(-1)
    idxs = np.ascontiguousarray(idxs, dtype=np.int64).reshape(-1)
    bv = ctypes.c_double(0.0)
    bi = ctypes.c_int64(-1)
    _lib.check(_lib.load().cbo_argmax_pairs(_lib.dptr(vals), idxs.ctypes.data_as(_lib.c_int64_p), vals.shape[0],
                                            ctypes.byref(bv), ctypes.byref(bi)))
    return bv.value, bi.value


NO_CANDIDATE = np.iinfo(np.int64).max     # index sent by a rank whose shard is empty


_exchange_buffers = {}     # (device, world size) -> persistent tensors of the per-step exchange


def _buffers(device, world):
    """Persistent pinned/device tensors and, on a GPU, a side stream of torch's own: the exchange then allocates
    nothing per step and never touches the legacy default stream (whose operations synchronise with every blocking
    stream of the process, the library's CU-masked ones included)."""
    import torch
    key = (str(device), world)
    buf = _exchange_buffers.get(key)
    if buf is None:
        on_gpu = device.type == "cuda"
        buf = {
            "host_in": torch.empty(2, dtype=torch.int64, pin_memory=on_gpu),
            "host_out": torch.empty(2 * world, dtype=torch.int64, pin_memory=on_gpu),
            "mine": torch.empty(2, dtype=torch.int64, device=device),
            "out": torch.empty(2 * world, dtype=torch.int64, device=device),
            "stream": torch.cuda.Stream(device=device) if on_gpu else None,
        }
        _exchange_buffers[key] = buf
    return buf


def exchange_argmax(best_val, best_idx, group=None, device=None):
    """All-gather this rank's (best_val, best_global_idx) and reduce.  Returns (val, idx) identical on
    every rank.  Without an initialised process group (plain single-GPU run) it is the identity; a group of
    one rank still goes through the collective."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(best_val), int(best_idx)
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
            else torch.device("cpu")
    device = torch.device(device)
    buf = _buffers(device, world)
    # one 16-byte record per rank: the value's bits and the index, both as int64 (one collective)
    buf["host_in"][0] = np.float64(best_val).view(np.int64).item()
    buf["host_in"][1] = int(best_idx)
    if buf["stream"] is not None:
        with torch.cuda.stream(buf["stream"]):
            buf["mine"].copy_(buf["host_in"], non_blocking=True)
            dist.all_gather_into_tensor(buf["out"], buf["mine"], group=group)
            buf["host_out"].copy_(buf["out"], non_blocking=True)
        buf["stream"].synchronize()
    else:
        buf["mine"].copy_(buf["host_in"])
        dist.all_gather_into_tensor(buf["out"], buf["mine"], group=group)
        buf["host_out"].copy_(buf["out"])
    rec = buf["host_out"].numpy().reshape(world, 2)
    vals = rec[:, 0].copy().view(np.float64)
    idxs = rec[:, 1].copy()
    keep = idxs != NO_CANDIDATE
    return reduce_pairs(vals[keep], idxs[keep])


def sharded_sweep(local_sweep, m_total, world_size, rank, group=None, device=None):
    """Run ``local_sweep(begin, end) -> (best_val, best_global_idx)`` on this rank's block and agree on
    the global winner.  ``local_sweep`` is the HIP sweep in the product (bench.py, CBO path); tests
    inject other callables to exercise the exchange on CPU ranks."""
    begin, end = shard_bounds(m_total, world_size, rank)
    if end > begin:
        val, idx = local_sweep(begin, end)
    else:
        val, idx = -np.inf, NO_CANDIDATE
    return exchange_argmax(val, idx, group=group, device=device)
