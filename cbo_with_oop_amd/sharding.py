"""Candidate-grid sharding across the GPUs of one node -- no PyTorch.

The acquisition sweep is independent per candidate once the posterior is fitted, so the grid is cut into contiguous
blocks, one per rank (one process per GPU); every rank fits the same posterior from the same (X, y) (replicated,
bit-identical, no data-path collective) and sweeps its block.  The only exchange is the arg-max: each rank
contributes (best_val, best_global_idx) = 16 bytes, all-gathered over RCCL (xGMI) by ``libcbo_hip.so`` itself
(``cbo_comm_*`` of include/cbo_hip.h: ``librccl.so.1`` is dlopen'ed there) and reduced with the same tie rule on
every rank (lowest global index wins; NaN maximal), so all ranks agree on the winner.  RCCL has no MAXLOC, hence
gather + local reduce.

Launch layout: the one ``torch.distributed.run`` / ``mpirun`` produce -- RANK, WORLD_SIZE, LOCAL_RANK in the
environment.  The 128-byte RCCL id travels from rank 0 to the others through a file in the temporary directory
(single node; a private file keyed by MASTER_PORT, the launcher's pid and run id, the user and the call's number) -- the launcher's own TCP store belongs to
the launcher.

Exploration sets are a second independent axis (src/CBO.py:249 loops over them): ``CBOAcquisitionPath`` places whole
sets on ranks when there are at least as many sets as ranks, and candidate blocks otherwise.
"""
from __future__ import annotations

import ctypes
import os
import tempfile
import time

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
ERROR_CANDIDATE = NO_CANDIDATE - 1        # index sent (with a NaN value, which wins every reduction) by a rank that failed
ID_BYTES = 128
_rendezvous_count = [0]                   # from_env calls made by this process: every rank counts the same calls


def _id_path(generation=0):
    """The rendezvous file of this launch: temp dir, MASTER_PORT, the launcher's run id and pid, and the number of the
    from_env call (a second communicator of the same run never reads the first one's id)."""
    tag = "_".join(str(os.environ.get(k, "x")) for k in ("MASTER_PORT", "TORCHELASTIC_RUN_ID"))
    return os.path.join(tempfile.gettempdir(), f"cbo_comm_{tag}_{os.getppid()}_{os.getuid()}_{int(generation)}.id")


def _publish_id(path, uid):
    """Rank 0: the 128 bytes appear atomically (O_EXCL temporary, mode 0600, rename); a leftover of an earlier run that
    crashed before its unlink is replaced."""
    tmp = f"{path}.{os.getpid()}.tmp"
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    try:
        os.write(fd, uid)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read_id(path):
    """Other ranks: only a regular file of exactly 128 bytes that this user owns and nobody else can write."""
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    try:
        st = os.fstat(fd)
        import stat
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022) or st.st_size != ID_BYTES:
            raise _lib.CboHipError(_lib.CBO_ERR_COMM, f"refusing the communicator id file {path}: not a private 128-byte file of this user")
        return os.read(fd, ID_BYTES)
    finally:
        os.close(fd)


class Communicator:
    """One rank of an RCCL communicator (``cbo_comm``).  ``from_env`` forms it from the launcher's environment;
    ``single`` forms a one-rank communicator (exercises the collective on a one-GPU box)."""

    def __init__(self, context, world, rank, id_bytes):
        self._lib = _lib.load()
        self._ctx = context
        self.world, self.rank = int(world), int(rank)
        buf = (ctypes.c_char * ID_BYTES).from_buffer_copy(id_bytes)
        self._handle = ctypes.c_void_p()
        _lib.check(self._lib.cbo_comm_init_rank(context.handle, self.world, self.rank, buf, ctypes.byref(self._handle)))

    @staticmethod
    def unique_id():
        buf = (ctypes.c_char * ID_BYTES)()
        _lib.check(_lib.load().cbo_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def single(cls, context):
        return cls(context, 1, 0, cls.unique_id())

    @classmethod
    def from_env(cls, context=None, timeout_s=120.0):
        """The communicator of this process under a one-process-per-GPU launcher, or None for a plain run
        (no RANK / WORLD_SIZE in the environment)."""
        if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
            return None
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        ctx = context if context is not None else _lib.Context.get()
        _rendezvous_count[0] += 1
        path = _id_path(_rendezvous_count[0])
        if rank == 0:
            uid = cls.unique_id()
            _publish_id(path, uid)                      # atomic: a reader sees all 128 bytes or no file
        else:
            deadline = time.monotonic() + timeout_s
            while not os.path.exists(path):
                if time.monotonic() > deadline:
                    raise _lib.CboHipError(_lib.CBO_ERR_COMM, f"rank {rank}: no communicator id at {path}")
                time.sleep(0.01)
            uid = _read_id(path)
        comm = cls(ctx, world, rank, uid)
        comm.barrier()                                   # every rank has read the id
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    def size(self):
        """(world, rank) as the RCCL communicator itself reports them (``cbo_comm_size``)."""
        w, r = ctypes.c_int(0), ctypes.c_int(-1)
        _lib.check(self._lib.cbo_comm_size(self._handle, ctypes.byref(w), ctypes.byref(r)))
        return w.value, r.value

    def argmax(self, best_val, best_idx):
        bv, bi = ctypes.c_double(0.0), ctypes.c_int64(-1)
        _lib.check(self._lib.cbo_comm_argmax(self._handle, float(best_val), int(best_idx), ctypes.byref(bv),
                                             ctypes.byref(bi)))
        return bv.value, bi.value

    def max(self, value):
        out = ctypes.c_double(0.0)
        _lib.check(self._lib.cbo_comm_max_f64(self._handle, float(value), ctypes.byref(out)))
        return out.value

    def barrier(self):
        _lib.check(self._lib.cbo_comm_barrier(self._handle))

    def gather(self, value):
        """One integer from every rank, in rank order (``cbo_comm_gather_i64``)."""
        out = (ctypes.c_int64 * self.world)()
        _lib.check(self._lib.cbo_comm_gather_i64(self._handle, int(value), out))
        return list(out)

    def share_factor(self, model, level, owners, needers):
        """The ranks of ``needers`` receive ``model``'s factor at ``level`` of the jitchol ladder from the ranks of
        ``owners``, one row slice from each (``cbo_comm_share_factor``); every rank calls this with the same lists."""
        own = (ctypes.c_int * max(1, len(owners)))(*owners)
        need = (ctypes.c_int * max(1, len(needers)))(*needers)
        _lib.check(self._lib.cbo_comm_share_factor(self._handle, model._handle, int(level), own, len(owners), need,
                                                   len(needers)))
        if self.rank in needers:
            model.adopted_factor(level)

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            if not self._ctx.closed:
                self._lib.cbo_comm_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_communicator(context=None):
    """The process-wide communicator formed from the environment on first use (None for a plain run)."""
    if "comm" not in _default:
        _default["comm"] = Communicator.from_env(context)
    return _default["comm"]


def exchange_argmax(best_val, best_idx, comm=None):
    """All-gather this rank's (best_val, best_global_idx) and reduce.  Returns (val, idx), identical on every
    rank.  Without a communicator (plain single-GPU run) it is the identity."""
    if comm is None:
        return float(best_val), int(best_idx)
    return comm.argmax(best_val, best_idx)


def sharded_sweep(local_sweep, m_total, world_size, rank, exchange=exchange_argmax):
    """Run ``local_sweep(begin, end) -> (best_val, best_global_idx)`` on this rank's block and agree on
    the global winner through ``exchange(val, idx)``.  ``local_sweep`` is the HIP sweep in the product (bench.py,
    CBO path); tests inject other callables (and a gloo exchange) to rehearse the logic on CPU ranks."""
    begin, end = shard_bounds(m_total, world_size, rank)
    if end > begin:
        val, idx = local_sweep(begin, end)
    else:
        val, idx = -np.inf, NO_CANDIDATE
    return exchange(val, idx)


# ---- jitchol's ladder walked by the ranks side by side --------------------------------------------------------------
# GPy's util.linalg.jitchol (behind GPRegression, /root/reference/src/GaussianProcessFactory.py:57-73) tries the plain
# factorisation, then mean(diag) * 1e-6 of jitter, x10 per retry, five retries at most, and keeps the FIRST level that
# goes through.  With the posterior replicated on G ranks (configs 3-5: candidate shards, one model) every rank repeats
# that walk: at config 4 a 26 ms attempt that fails, then the 31 ms one that succeeds, on all eight GPUs -- time that
# does not shrink with G.  Side by side instead: every level below the one expected to succeed gets ONE rank (a
# verifier: the sequential walk's answer needs those levels to have failed), every other rank tries the expected level
# (replicas: no transfer needed among them); one small all-gather of the outcomes later every rank knows the lowest
# level that went through -- the sequential walk's answer, exactly -- and the verifiers receive the factor from the
# replicas, one row slice from each (|replicas| xGMI links at once).  The expected level is the one the model's last fit
# needed.  When the expectation is wrong the protocol still ends with the sequential answer: a verifier's level went
# through (it is the lowest: everybody else receives from it), or nothing did (the next round starts above the levels
# tried).  One rank: the sequential walk.
LADDER_LAST_LEVEL = 5          # jitchol: the plain attempt (level 0) and five retries


def ladder_plan(world, first_level, expected_level):
    """Level every rank tries this round: the levels ``first_level .. expected_level - 1`` one verifier rank each, all
    remaining ranks ``expected_level``; fewer ranks than that needs: consecutive levels from ``first_level``."""
    expected_level = max(expected_level, first_level)
    verifiers = expected_level - first_level
    if world > verifiers:
        return [first_level + r if r < verifiers else expected_level for r in range(world)]
    return [first_level + r for r in range(world)]


def ladder_resolve(levels, outcomes):
    """From every rank's level and outcome (1 factored, 0 not positive definite, -1 non-positive diagonal, None: level
    beyond the ladder, not tried): ``(level, owners, needers, next_first_level)`` -- the lowest level that went through
    with the ranks that hold / lack its factor, or ``level`` None and where the next round starts.  The levels of a
    round are consecutive from its first (``ladder_plan``), so the lowest success is the sequential walk's answer.
    Raises ``numpy.linalg.LinAlgError`` as jitchol does (on every rank alike)."""
    good = sorted({lv for lv, ok in zip(levels, outcomes) if ok == 1})
    if good:
        level = good[0]
        owners = [r for r, lv in enumerate(levels) if lv == level]
        if any(outcomes[r] != 1 for r in owners):
            raise RuntimeError(f"ranks that tried level {level} disagree: {[outcomes[r] for r in owners]}")
        return level, owners, [r for r, lv in enumerate(levels) if lv != level], None
    if any(ok == -1 for ok in outcomes):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    nxt = max(levels) + 1
    if nxt > LADDER_LAST_LEVEL:
        raise np.linalg.LinAlgError("not positive definite, even with jitter.")
    return None, [], [], nxt


def fit_over_ranks(model, comm=None, expected_level=None):
    """Fit ``model`` (replicated on every rank of ``comm``) with jitchol's ladder walked side by side; returns
    ``(level, jitter)`` -- what ``model.jitter_tries, model.jitter`` are after a plain fit.  ``comm`` needs ``world``,
    ``rank``, ``gather(int) -> list`` and ``share_factor(model, level, owners, needers)`` (sharding.Communicator over
    RCCL; the tests drive the same code over threads and over gloo); ``model`` needs ``fit_level(level) -> (outcome,
    jitter)`` and ``adopted_factor(level)`` (HipGaussianProcess)."""
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    first = 0
    expected = int(getattr(model, "jitter_tries", 0) or 0) if expected_level is None else int(expected_level)
    while True:
        levels = ladder_plan(world, first, min(expected, LADDER_LAST_LEVEL))
        mine = levels[rank]
        # A rank whose attempt raises (a HIP error, a level beyond the ladder) must not leave the others in the gather: its
        # error travels as an outcome of its own (-3), and every rank raises together.
        failure = None
        try:
            outcome, jitter = (None, 0.0) if mine > LADDER_LAST_LEVEL else model.fit_level(mine)
        except Exception as exc:                     # noqa: BLE001 -- whatever it is, the peers have to hear of it
            if comm is None:
                raise
            outcome, failure = -3, exc
        outcomes = comm.gather(-2 if outcome is None else outcome) if comm is not None else [outcome]
        if any(o == -3 for o in outcomes):
            if failure is not None:
                raise failure
            raise RuntimeError(f"fit_over_ranks: the fit of rank(s) {[r for r, o in enumerate(outcomes) if o == -3]} failed "
                               f"(their own exception says why); this rank stops with them")
        outcomes = [None if o == -2 else o for o in outcomes]
        level, owners, needers, first = ladder_resolve(levels, outcomes)
        if level is not None:
            if needers:
                comm.share_factor(model, level, owners, needers)
            return level, model.jitter
        expected = first


def factor_slices(n_pad, n_owners):
    """Rows ``[begin, end)`` of the factor that owner number i sends (``cbo_comm_share_factor``'s split: whole 128-row
    blocks, the first slices one block longer when the blocks do not divide)."""
    blocks = n_pad // 128
    base, extra = divmod(blocks, n_owners)
    out, b0 = [], 0
    for i in range(n_owners):
        b1 = b0 + base + (1 if i < extra else 0)
        out.append((128 * b0, 128 * b1))
        b0 = b1
    return out
