"""world_size-2 (and 3) `gloo` rehearsal of the multi-GPU path on CPU ranks: candidate sharding and the arg-max
exchange logic of cbo_with_oop_amd/sharding.py.  The product's exchange is RCCL inside libcbo_hip.so (cbo_comm_*,
needs GPUs: tests/test_parity_gpu.py drives it); here the all-gather is torch.distributed/gloo, defined in this test,
and the per-shard scores come from the oracle (stand-in for the HIP sweep).  Under test: the partition, the empty-shard
sentinel and the tie rule (the library's own cbo_argmax_pairs reduces the gathered records)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_fixture


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from cbo_with_oop_amd.sharding import NO_CANDIDATE, reduce_pairs, sharded_sweep
    from oracle import gp_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def gloo_exchange(val, idx):
        # one 16-byte record per rank (value bits, index), gathered; reduced by the library's host reduction
        mine = torch.tensor([np.float64(val).view(np.int64).item(), int(idx)], dtype=torch.int64)
        out = torch.empty(2 * world, dtype=torch.int64)
        dist.all_gather_into_tensor(out, mine)
        rec = out.numpy().reshape(world, 2)
        vals, idxs = rec[:, 0].copy().view(np.float64), rec[:, 1].copy()
        keep = idxs != NO_CANDIDATE
        return reduce_pairs(vals[keep], idxs[keep])

    f = load_fixture(name)
    post = O.fit(f["X"], f["y"], f["mX"], f["vX"], float(f["variance"]), f["lengthscale_arg"], float(f["noise_var"]))

    def local(begin, end):
        sl = slice(begin, end)
        mXs = None if f["mXs"] is None else f["mXs"][sl]
        vXs = None if f["vXs"] is None else f["vXs"][sl]
        acq, val, idx, _, _ = O.acquisition_sweep(post, f["Xs"][sl], float(f["y_best"]), mXs, vXs, f["task"],
                                                  float(f["cost"]))
        return val, begin + idx

    val, idx = sharded_sweep(local, f["Xs"].shape[0], world, rank, exchange=gloo_exchange)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([val, idx]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "toy_bo_d2"), (2, "coral_max_d3"), (3, "causal_d2")])
def test_sharded_argmax_equals_unsharded(tmp_path, world, name):
    f = load_fixture(name)
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    for r in res:                                   # every rank agrees, and agrees with the unsharded sweep
        assert int(r[1]) == int(f["best_idx"])
        assert r[0] == float(f["best_val"]) or (r[0] == 0 and float(f["best_val"]) == 0)


def test_more_ranks_than_candidates(tmp_path):
    """Empty shards must not win (their index is the NO_CANDIDATE sentinel)."""
    from cbo_with_oop_amd.sharding import shard_bounds
    assert shard_bounds(2, 4, 3) == (2, 2)


def test_rendezvous_file_is_private_atomic_and_per_call(tmp_path, monkeypatch):
    """The id file of Communicator.from_env: created exclusively with mode 0600 and renamed into place, a different name
    for every from_env call of a run, and a reader refuses anything that is not a private 128-byte file of this user."""
    import stat
    from cbo_with_oop_amd import _lib, sharding
    monkeypatch.setattr(sharding.tempfile, "gettempdir", lambda: str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29511")
    p1, p2 = sharding._id_path(1), sharding._id_path(2)
    assert p1 != p2 and str(os.getppid()) in p1 and p1.startswith(str(tmp_path))
    uid = bytes(range(128))
    sharding._publish_id(p1, uid)
    assert stat.S_IMODE(os.stat(p1).st_mode) == 0o600 and sharding._read_id(p1) == uid
    sharding._publish_id(p1, uid[::-1])                                  # a leftover is replaced, atomically
    assert sharding._read_id(p1) == uid[::-1]
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    with open(p2, "wb") as f:
        f.write(b"short")
    with pytest.raises(_lib.CboHipError):
        sharding._read_id(p2)
    os.chmod(p1, 0o666)
    with pytest.raises(_lib.CboHipError):
        sharding._read_id(p1)


def test_a_failed_rank_wins_every_reduction_with_its_error_record():
    """A rank that failed contributes (NaN, ERROR_CANDIDATE): NaN is maximal in the library's reduction, so every rank
    sees the error index and raises instead of waiting for a record that never comes."""
    from cbo_with_oop_amd.sharding import ERROR_CANDIDATE, NO_CANDIDATE, reduce_pairs
    val, idx = reduce_pairs([0.3, float("nan"), 0.9], [5, ERROR_CANDIDATE, 77])
    assert idx == ERROR_CANDIDATE and np.isnan(val)
    assert ERROR_CANDIDATE != NO_CANDIDATE


def _ladder_worker(rank, world, port, needs, expected, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from cbo_with_oop_amd.sharding import factor_slices, fit_over_ranks
    from ladder_support import LadderModel
    from test_ladder_ranks import matrix_needing
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooComm:
        """gather / share_factor of sharding.Communicator over gloo: an all-gather of one int64, and point-to-point
        sends of the factor's row slices (cbo_comm_share_factor's split: one slice from every owner to every needer)."""
        def __init__(self):
            self.world, self.rank, self.received = world, rank, []

        def gather(self, value):
            out = torch.empty(world, dtype=torch.int64)
            dist.all_gather_into_tensor(out, torch.tensor([int(value)], dtype=torch.int64))
            return out.tolist()

        def share_factor(self, model, level, owners, needers):
            n = model.A.shape[0]
            n_pad = -(-n // 128) * 128
            slices = factor_slices(n_pad, len(owners))
            if rank in owners:
                r0, r1 = slices[owners.index(rank)]
                block = torch.from_numpy(np.ascontiguousarray(model.L[r0:min(r1, n)]))
                for dst in needers:
                    if block.numel():
                        dist.send(block, dst)
            elif rank in needers:
                L = np.zeros((n, n))
                for (r0, r1), src in zip(slices, owners):
                    block = torch.empty((max(0, min(r1, n) - r0), n), dtype=torch.float64)
                    if block.numel():
                        dist.recv(block, src)
                        L[r0:min(r1, n)] = block.numpy()
                        self.received.append(src)
                model.L = L
                model.adopted_factor(level)

    comm = GlooComm()
    model = LadderModel(matrix_needing(needs, seed=needs))
    level, jitter = fit_over_ranks(model, comm, expected)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), L=model.L, level=level, jitter=jitter, tried=np.array(model.tried),
             received=np.array(comm.received, dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,needs,expected", [(2, 1, 1), (2, 2, None), (3, 1, 1), (3, 1, 2)])
def test_ladder_walked_by_gloo_ranks_equals_the_sequential_walk(tmp_path, world, needs, expected):
    """sharding.fit_over_ranks over real processes (gloo): every rank ends with the oracle's jitchol result; when the
    expected level holds, rank 0 only tries the plain factorisation and receives the factor from the other ranks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import gp_oracle as O
    from test_ladder_ranks import matrix_needing
    mp.spawn(_ladder_worker, args=(world, _free_port(), needs, expected, str(tmp_path)), nprocs=world, join=True)
    L_ref, jit_ref, tries_ref = O.jitchol(matrix_needing(needs, seed=needs))
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in res:
        assert int(r["level"]) == tries_ref and float(r["jitter"]) == jit_ref
        assert np.array_equal(r["L"], L_ref)
    if expected == needs and world > needs:
        assert res[0]["tried"].tolist() == [0] and sorted(res[0]["received"].tolist()) == list(range(needs, world))
        assert all(r["tried"].tolist() == [needs] for r in res[needs:])
