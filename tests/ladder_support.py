"""Stand-ins for the device model and the communicator of sharding.fit_over_ranks (jitchol's ladder walked by the ranks
side by side), for the CPU tests: the model factors with LAPACK exactly as the oracle's jitchol does, level by level;
the communicators move numpy arrays between threads (ThreadComm) or gloo ranks (tests/test_sharding_gloo.py)."""
import threading

import numpy as np
from scipy.linalg import lapack

from cbo_with_oop_amd.sharding import factor_slices


class LadderModel:
    """``fit_level`` / ``adopted_factor`` of HipGaussianProcess over a host matrix: level 0 = dpotrf(A), level k =
    dpotrf(A + mean(diag) * 1e-6 * 10^(k-1) I) -- the oracle's (GPy's) jitchol, one level at a time."""

    def __init__(self, A, last_level=0):
        self.A = np.ascontiguousarray(A, dtype=np.float64)
        self.jitter_tries, self.jitter = last_level, 0.0
        self.L = None
        self.tried = []

    def level_jitter(self, level):
        j = 0.0
        for k in range(level):
            j = np.diag(self.A).mean() * 1e-6 if k == 0 else j * 10
        return j

    def fit_level(self, level):
        self.tried.append(level)
        self.L = None
        if level >= 1 and np.any(np.diag(self.A) <= 0.0):
            return -1, 0.0
        j = self.level_jitter(level)
        L, info = lapack.dpotrf(np.ascontiguousarray(self.A + np.eye(self.A.shape[0]) * j) if level else self.A, lower=1)
        if info != 0:
            return 0, j
        self.L, self.jitter_tries, self.jitter = np.tril(L), level, j
        return 1, j

    def adopted_factor(self, level):
        self.jitter_tries, self.jitter = level, self.level_jitter(level)


class ThreadComm:
    """world ranks = world threads of this process; ``gather`` and ``share_factor`` with the semantics of
    sharding.Communicator (cbo_comm_gather_i64 / cbo_comm_share_factor: one row slice from every owner)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.models = [None] * world
            self.transfers = []

    def __init__(self, shared, rank):
        self.shared, self.world, self.rank = shared, shared.world, rank

    def gather(self, value):
        self.shared.slots[self.rank] = int(value)
        self.shared.barrier.wait()
        out = list(self.shared.slots)
        self.shared.barrier.wait()
        return out

    def share_factor(self, model, level, owners, needers):
        sh = self.shared
        sh.models[self.rank] = model
        sh.barrier.wait()
        if self.rank in needers:
            n = model.A.shape[0]
            n_pad = -(-n // 128) * 128
            L = np.zeros((n_pad, n))
            for (r0, r1), owner in zip(factor_slices(n_pad, len(owners)), owners):
                src = sh.models[owner].L
                L[r0:min(r1, n)] = src[r0:min(r1, n)]
                sh.transfers.append((owner, self.rank, r0, r1))
            model.L = L[:n]
            model.adopted_factor(level)
        sh.barrier.wait()


def run_ranks(world, make_model, expected_level=None):
    """fit_over_ranks on `world` threads; returns the models and the shared state (or raises what a rank raised)."""
    from cbo_with_oop_amd.sharding import fit_over_ranks
    shared = ThreadComm.Shared(world)
    models = [make_model(r) for r in range(world)]
    results, errors = [None] * world, [None] * world

    def body(r):
        try:
            results[r] = fit_over_ranks(models[r], ThreadComm(shared, r) if world > 1 else None, expected_level)
        except np.linalg.LinAlgError as e:         # jitchol's verdict: every rank reaches it on its own
            errors[r] = e
        except Exception as e:                     # noqa: BLE001 -- handed to the test; the other ranks must not hang
            errors[r] = e
            shared.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    return models, results, errors, shared
