"""jitchol's ladder walked by the ranks side by side (cbo_with_oop_amd/sharding.py: ladder_plan, ladder_resolve,
fit_over_ranks): whatever the number of ranks and whatever level they expect, every rank ends with the factor, the
retry count and the jitter of the sequential walk -- GPy's util.linalg.jitchol as restated in oracle/gp_oracle.py
(reference call site: /root/reference/src/GaussianProcessFactory.py:57-73).  Ranks are threads here; the transport over
RCCL is cbo_comm_gather_i64 / cbo_comm_share_factor (tests/test_parity_gpu.py drives them on one rank), over gloo
tests/test_sharding_gloo.py."""
import numpy as np
import pytest

from ladder_support import LadderModel, run_ranks
from cbo_with_oop_amd.sharding import LADDER_LAST_LEVEL, factor_slices, ladder_plan, ladder_resolve
from oracle import gp_oracle as O


def matrix_needing(level, n=200, seed=0):
    """A symmetric matrix whose plain factorisation fails and whose jitchol ladder first succeeds at `level` (0: positive
    definite as it is): a Gram matrix of rank n/2 shifted down by a fraction of the level's jitter."""
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, n // 2))
    A = B @ B.T
    A = 0.5 * (A + A.T)
    if level == 0:
        return A + np.eye(n)
    mean = np.diag(A).mean()
    shift = mean * 1e-6 * 10.0 ** (level - 1) * 0.3             # needs more than a tenth of this level's jitter
    A = A - np.eye(n) * shift
    L, jit, tries = O.jitchol(A)
    assert tries == level, (tries, level)
    return A


def test_plan_gives_every_level_below_the_expected_one_a_verifier():
    assert ladder_plan(1, 0, 0) == [0]
    assert ladder_plan(1, 0, 3) == [0]                          # one rank: the sequential walk
    assert ladder_plan(8, 0, 0) == [0] * 8
    assert ladder_plan(8, 0, 1) == [0, 1, 1, 1, 1, 1, 1, 1]
    assert ladder_plan(8, 0, 3) == [0, 1, 2, 3, 3, 3, 3, 3]
    assert ladder_plan(2, 0, 3) == [0, 1]                       # too few ranks for replicas: consecutive levels
    assert ladder_plan(4, 2, 1) == [2, 2, 2, 2]                 # a later round never goes back below its first level
    assert ladder_plan(3, 4, 5) == [4, 5, 5]


def test_resolve_is_the_sequential_answer():
    assert ladder_resolve([0, 1, 1, 1], [0, 1, 1, 1]) == (1, [1, 2, 3], [0], None)
    assert ladder_resolve([0, 1, 2, 2], [0, 1, 1, 1]) == (1, [1], [0, 2, 3], None)      # the expectation was too high
    assert ladder_resolve([0, 0], [1, 1]) == (0, [0, 1], [], None)
    assert ladder_resolve([0, 1], [0, 0]) == (None, [], [], 2)
    with pytest.raises(np.linalg.LinAlgError, match="even with jitter"):
        ladder_resolve([4, 5, 5], [0, 0, 0])
    with pytest.raises(np.linalg.LinAlgError, match="non-positive diagonal"):
        ladder_resolve([0, 1, 1], [0, -1, -1])
    with pytest.raises(RuntimeError):
        ladder_resolve([1, 1], [1, 0])                          # replicas of one level cannot disagree


def test_factor_slices_cover_the_rows_in_whole_blocks():
    for n_pad in (128, 384, 4096, 16384):
        for owners in (1, 2, 3, 7):
            sl = factor_slices(n_pad, owners)
            assert sl[0][0] == 0 and sl[-1][1] == n_pad and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert all(b % 128 == 0 and e % 128 == 0 for b, e in sl)
            sizes = [e - b for b, e in sl]
            assert max(sizes) - min(sizes) <= 128


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("needs,expected", [(0, None), (1, None), (1, 1), (1, 3), (2, 1), (3, 3), (5, 5), (2, 0)])
def test_every_rank_ends_with_the_sequential_walk(world, needs, expected):
    A = matrix_needing(needs, seed=needs)
    L_ref, jit_ref, tries_ref = O.jitchol(A)
    models, results, errors, shared = run_ranks(world, lambda r: LadderModel(A), expected)
    assert errors == [None] * world, errors
    for m, res in zip(models, results):
        assert res == (tries_ref, jit_ref) and (m.jitter_tries, m.jitter) == (tries_ref, jit_ref)
        assert np.array_equal(m.L, L_ref)
    if world > 1 and expected is not None and expected == needs and needs >= 1 and world > needs:
        # the expectation held: one round, every level below it tried by exactly one rank, the rest replicas; the
        # verifiers received the factor, one slice from every replica
        assert [m.tried for m in models[:needs]] == [[lv] for lv in range(needs)]
        assert all(m.tried == [needs] for m in models[needs:])
        owners = set(range(needs, world))
        assert {t[1] for t in shared.transfers} == set(range(needs))
        for v in range(needs):
            assert {t[0] for t in shared.transfers if t[1] == v} == owners


@pytest.mark.parametrize("world", [1, 2, 8])
def test_a_matrix_the_ladder_cannot_rescue_raises_on_every_rank(world):
    rng = np.random.default_rng(1)
    B = rng.standard_normal((64, 64))
    A = 0.5 * (B + B.T)                                         # indefinite, positive diagonal forced below
    A[np.diag_indices(64)] = np.abs(A[np.diag_indices(64)]) + 0.1
    with pytest.raises(O.NotPositiveDefinite):
        O.jitchol(A)
    models, results, errors, _ = run_ranks(world, lambda r: LadderModel(A))
    assert all(isinstance(e, np.linalg.LinAlgError) and "even with jitter" in str(e) for e in errors), errors
    assert max(max(m.tried) for m in models) == LADDER_LAST_LEVEL


@pytest.mark.parametrize("world", [1, 3])
def test_non_positive_diagonal_is_reported_as_jitchol_reports_it(world):
    A = matrix_needing(1, n=64)
    A[5, 5] = -1.0
    with pytest.raises(O.NotPositiveDefinite, match="non-positive diagonal"):
        O.jitchol(A)
    _, _, errors, _ = run_ranks(world, lambda r: LadderModel(A), 1)
    assert all(isinstance(e, np.linalg.LinAlgError) and "non-positive diagonal" in str(e) for e in errors), errors
