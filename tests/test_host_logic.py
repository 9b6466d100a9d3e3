"""CPU tests of the host-side mirror of the reference interface (no device calls)."""
import os

import numpy as np
import pytest

from cbo_with_oop_amd.CBO import CBOAcquisitionPath
from cbo_with_oop_amd.GaussianProcessFactory import GaussianProcessType
from cbo_with_oop_amd.graphs import GRAPHS, CompleteGraph, CoralGraph, ToyGraph, meshgrid_candidates
from cbo_with_oop_amd.sharding import reduce_pairs, shard_bounds
from cbo_with_oop_amd.utils_functions.cost_functions import Cost, total_cost
from cbo_with_oop_amd.utils_functions.utils import default_grid_shape, find_current_global, space_bounds
from conftest import ROOT
from oracle import gp_oracle as O


def test_enum_values_match_reference():
    assert (GaussianProcessType.GRAPH_GP, GaussianProcessType.CAUSAL_GP, GaussianProcessType.NON_CAUSAL_GP) == (0, 1, 2)


def test_graph_tables():
    assert CompleteGraph.get_exploration_set("MIS") == [['B'], ['D'], ['E'], ['B', 'D'], ['B', 'E'], ['D', 'E']]
    assert len(CompleteGraph.get_exploration_set("POMIS")) == 5
    assert len(CoralGraph.get_exploration_set("MIS")) == 25
    assert CoralGraph.get_interventional_ranges()["T"] == [2450, 2500]
    assert ToyGraph.bounds(["X", "Z"]) == [(-5, 5), (-5, 20)]
    assert set(GRAPHS) == {"complete_graph", "coral_graph", "simplified_coral_graph", "toy_graph"}
    with pytest.raises(RuntimeError):
        CompleteGraph.get_cost_structure(5)


@pytest.mark.parametrize("type_cost", [1, 2, 3, 4])
def test_cost_matches_oracle(type_cost):
    costs = CompleteGraph.get_cost_structure(type_cost)
    x = np.array([[1.0, -2.0], [3.0, 4.0], [-0.5, 0.25]])
    es = ["B", "E"]
    fixed = {1: [1, 1], 2: [10, 20], 3: [10, 20], 4: [1, 1]}[type_cost]
    variable = type_cost in (3, 4)
    assert Cost(costs, es).evaluate(x) == O.cost_of_batch(x, fixed, [variable] * 2)
    assert total_cost(es, costs, {"B": 2.0, "E": -3.0}) == sum(fixed) + (5.0 if variable else 0.0)


def test_meshgrid_order_and_grid_shapes():
    g = meshgrid_candidates([(0, 1), (10, 12)], [2, 3])
    assert np.array_equal(g, [[0, 10], [0, 11], [0, 12], [1, 10], [1, 11], [1, 12]])     # first dim slowest
    assert default_grid_shape(1) == [200] and default_grid_shape(3) == [32, 32, 16]
    assert np.prod(default_grid_shape(2)) <= 16384

    class P:
        def __init__(self, lo, hi): self.min, self.max = lo, hi

    class S:
        parameters = [P(-1, 2), P(0, 3)]
    assert space_bounds(S()) == [(-1, 2), (0, 3)]
    assert space_bounds([(0, 1)]) == [(0, 1)]


def test_find_current_global_and_selection():
    cur = {"X": [np.inf, -1.0, -2.5], "Z": [np.inf]}
    assert find_current_global(cur, ["X", "Z"], "min") == O.find_current_global(cur, ["X", "Z"], "min") == -2.5
    cur = {"X": [-np.inf, 3.0], "Z": [-np.inf, 4.0]}
    assert find_current_global(cur, ["X", "Z"], "max") == 4.0
    path = CBOAcquisitionPath(GaussianProcessType.NON_CAUSAL_GP, [["X"], ["Z"], ["X", "Z"]], {}, "min",
                              [None] * 3, [None] * 3, [None] * 3)
    ys = [np.array([[0.2]]), np.array([[0.9]]), np.array([[0.9]])]
    assert path.select_next_intervention(ys) == (["Z"], 1)
    assert path.select_intervention(ys) == (["Z"], 1) and path.last_intervention == 1
    assert O.select_next_intervention(ys) == 1


def test_shard_bounds_cover_the_grid():
    for m in (1, 7, 64, 16384, 100003):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_bounds(m, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == m
            assert all(b[1] == nb[0] for b, nb in zip(blocks, blocks[1:]))
    assert reduce_pairs([0.5, 0.9, 0.9], [5, 900, 300]) == (0.9, 300)
    assert reduce_pairs([np.nan, 0.9], [7, 3]) [1] == 7          # NaN is maximal, like numpy.argmax


def test_one_row_growth_detection():
    """Host logic of the append shortcut: set_data only takes it when the new data are the resident ones plus exactly
    one observation and the model is fitted (no device involved: the method only reads attributes)."""
    from types import SimpleNamespace
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    X = np.arange(12.0).reshape(6, 2)
    Y = np.arange(6.0).reshape(6, 1)
    model = SimpleNamespace(stale=False, X=X, Y=Y)
    grew = lambda x, y: HipGaussianProcess._grew_by_one_row(model, x, y)
    x7, y7 = np.vstack([X, [[1.0, 2.0]]]), np.vstack([Y, [[3.0]]])
    assert grew(x7, y7) and grew(x7, y7[:, 0])
    assert not grew(X, Y)                                         # same data
    assert not grew(np.vstack([x7, [[0.0, 0.0]]]), np.vstack([y7, [[0.0]]]))   # two new rows
    changed = x7.copy(); changed[2, 0] += 1e-12
    assert not grew(changed, y7)                                  # an old row changed
    ychanged = y7.copy(); ychanged[0, 0] = -1.0
    assert not grew(x7, ychanged)
    assert not grew(np.hstack([x7, x7[:, :1]]), y7)               # another dimension
    model.stale = True
    assert not grew(x7, y7)                                       # nothing fitted to extend


def test_missing_rccl_is_reported_as_comm_error():
    """The arg-max exchange opens librccl with dlopen inside libcbo_hip.so: when it cannot, the C-ABI says
    CBO_ERR_COMM (no GPU needed for that; a fresh process because the library is opened once per process)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cbo_with_oop_amd import _lib\n"
            "from cbo_with_oop_amd.sharding import Communicator\n"
            "try:\n    Communicator.unique_id()\nexcept _lib.CboHipError as e:\n    print('code', e.code)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CBO_HIP_RCCL_LIB="/nonexistent/librccl.so"),
                         capture_output=True, text=True, timeout=120)
    assert "code -8" in out.stdout, (out.stdout, out.stderr[-500:])


def test_lockstep_lbfgs_reaches_scipys_optima_in_one_call_per_round():
    """The multi-start refinement's optimiser on an analytic multi-modal function: every start ends at the optimum
    scipy's L-BFGS-B finds from the same start, inside the box, with one batched evaluation per round."""
    from scipy.optimize import fmin_l_bfgs_b
    from cbo_with_oop_amd.utils_functions.causal_optimizer import lockstep_lbfgs
    rng = np.random.default_rng(0)
    C, W = rng.uniform(-3, 3, (6, 2)), rng.uniform(0.5, 2, 6)
    shapes = []

    def fun(X):
        shapes.append(X.shape)
        d = X[:, None, :] - C[None]
        e = W * np.exp(-0.5 * (d ** 2).sum(-1))
        return e.sum(1), -(e[:, :, None] * d).sum(1)

    lo, hi = np.array([-3.0, -2.0]), np.array([3.0, 2.5])
    x0 = rng.uniform(lo, hi, (8, 2))
    X, F = lockstep_lbfgs(fun, x0, lo, hi)
    assert set(shapes) == {(8, 2)} and len(shapes) < 60
    assert np.all(X >= lo) and np.all(X <= hi)
    f0, _ = fun(x0)
    assert np.all(F >= f0)
    for i in range(8):
        xs, fs, _ = fmin_l_bfgs_b(lambda v: (-fun(v[None])[0][0], -fun(v[None])[1][0]), x0[i], bounds=list(zip(lo, hi)))
        assert np.isclose(F[i], -fs, rtol=1e-8), (i, F[i], -fs)
    # a start already at a bound-constrained optimum stays there
    Xb, Fb = lockstep_lbfgs(lambda X: (X[:, 0], np.tile([1.0, 0.0], (len(X), 1))), np.array([[3.0, 0.0]]), lo, hi)
    assert np.allclose(Xb, [[3.0, 0.0]]) and np.isclose(Fb[0], 3.0)


def test_uniform_anchor_mode_consumes_the_global_generator_as_the_reference_does():
    """``CausalGradientAcquisitionOptimizer(space, anchors="uniform")`` is the reference's optimiser
    (src/utils_functions/causal_optimizer.py:19,52-65): 100 anchors from numpy's GLOBAL generator, drawn parameter by
    parameter as emukit's ``ParameterSpace.sample_uniform`` draws them, ONE batched ``evaluate`` over all of them, the top
    one by ``argsort()[::-1][:1]``, L-BFGS (scipy ``fmin_l_bfgs_b``, bounds, maxfun=1000) from it, its result returned as it
    comes.  Checked against a scripted replay of those steps on a smooth stand-in acquisition (no device): the draws
    consumed, the anchor chosen, the state the generator is left in."""
    from scipy.optimize import fmin_l_bfgs_b
    from cbo_with_oop_amd.utils_functions.causal_optimizer import CausalGradientAcquisitionOptimizer
    bounds = [(-5.0, 4.0), (-5.0, 5.0), (-6.0, 3.0)]
    centre = np.array([1.0, -2.0, 0.5])

    class Acq:
        calls = []

        def evaluate(self, x):
            Acq.calls.append(np.array(x, copy=True))
            return np.exp(-0.5 * np.sum((x - centre) ** 2, axis=1, keepdims=True))

        def evaluate_with_gradients(self, x):
            f = np.exp(-0.5 * np.sum((x - centre) ** 2, axis=1, keepdims=True))
            return f, -f * (x - centre)

    np.random.seed(9)                                          # (--seed 9, src/ArgumentParser.py:24,47)
    opt = CausalGradientAcquisitionOptimizer(bounds, anchors="uniform")
    x, fx = opt.optimize(Acq())
    after = np.random.uniform()
    # the scripted replay
    np.random.seed(9)
    cols = [np.random.uniform(low=lo, high=hi, size=(100, 1)) for lo, hi in bounds]
    anchors = np.hstack(cols)
    assert np.random.uniform() == after                        # exactly 300 draws were consumed, nothing else
    assert len(Acq.calls) == 1 and np.array_equal(Acq.calls[0], anchors)       # ONE batched evaluate, over exactly these
    scores = np.exp(-0.5 * np.sum((anchors - centre) ** 2, axis=1))
    start = anchors[np.argsort(scores)[::-1][:1]]
    xr, fr, _ = fmin_l_bfgs_b(lambda v: (-float(np.exp(-0.5 * np.sum((v - centre) ** 2))),
                                         float(np.exp(-0.5 * np.sum((v - centre) ** 2))) * (v - centre)),
                              start.reshape(-1), bounds=bounds, maxfun=1000)
    assert np.array_equal(x, xr[None, :]) and fx[0, 0] == -fr and np.allclose(x, centre, atol=1e-4)
    # the default stays the grid; anything else is refused
    assert CausalGradientAcquisitionOptimizer(bounds).anchors == "grid"
    with pytest.raises(ValueError):
        CausalGradientAcquisitionOptimizer(bounds, anchors="sobol")


def test_bench_configs_are_the_baseline_shapes_and_shard_as_stated():
    """bench.py --config c2..c5: BASELINE.json's sizes, the reference's interventional ranges, weak scaling stacks one
    grid per rank, strong scaling cuts the ONE grid, and an 8-GPU config on fewer ranks keeps 1/8 per rank unless told."""
    import bench
    from cbo_with_oop_amd.graphs import CompleteGraph, CoralGraph, SimplifiedCoralGraph
    from cbo_with_oop_amd.sharding import shard_bounds
    c = bench.CONFIGS
    assert (c["c2"]["n_obs"], np.prod(c["c2"]["grid"])) == (4096, 16384)
    assert (c["c3"]["n_obs"], np.prod(c["c3"]["grid"])) == (8192, 65536)
    assert (c["c4"]["n_obs"], np.prod(c["c4"]["grid"])) == (16384, 262144) and c["c4"]["dtype"] == "f64"
    assert (c["c5"]["n_obs"], np.prod(c["c5"]["grid"])) == (16384, 262144) and c["c5"]["dtype"] == "f32"
    assert c["c3"]["box"] == [tuple(map(float, b)) for b in CompleteGraph.bounds(["B", "D", "E"])]
    assert c["c4"]["box"] == [tuple(map(float, b)) for b in SimplifiedCoralGraph.bounds(["N", "O", "T"])]
    assert c["c5"]["box"] == [tuple(map(float, b)) for b in CoralGraph.bounds(["N", "O", "T"])]
    small = dict(c["c3"], n_obs=64, grid=(8, 4, 4))                       # the same code path on a small stand-in
    X, y, Xs, grid, note = bench.make_problem(small, 4, "weak", False)
    assert X.shape == (64, 3) and Xs.shape[0] == 4 * 128 and grid == (8, 4, 16)
    X, y, Xs, grid, note = bench.make_problem(small, 4, "strong", False)
    assert Xs.shape[0] == 128 and grid == (8, 4, 4) and "4 contiguous shard" in note
    assert sum(e - b for b, e in (shard_bounds(128, 4, r) for r in range(4))) == 128
    eight = dict(c["c4"], n_obs=64, grid=(8, 8, 8))
    _, _, Xs1, _, note1 = bench.make_problem(eight, 1, "strong", False)
    _, _, Xs2, _, _ = bench.make_problem(eight, 2, "strong", False)
    _, _, Xs8, _, _ = bench.make_problem(eight, 8, "strong", False)
    _, _, Xsf, _, _ = bench.make_problem(eight, 1, "strong", True)
    assert (Xs1.shape[0], Xs2.shape[0], Xs8.shape[0], Xsf.shape[0]) == (64, 128, 512, 512) and "1/8" in note1
    assert np.array_equal(Xs1, Xs8[:64])                                   # rank 0's shard of the 8-GPU run


def _run_bench(args, extra_env, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


def test_bench_starts_its_own_ranks_for_a_plain_multi_gpu_invocation():
    """`python bench.py --gpus 2 --config c3` with no launcher around it (the form the driver uses): the process becomes
    the launcher, two rank processes are its children with the launcher's environment contract (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1, one MASTER_PORT), each takes its contiguous shard of the fixed grid
    (src/CBO.py:237-260 is the loop being sharded), rank 0's line is the launcher's standard output, and nothing has loaded
    the library (CBO_HIP_LIB points nowhere: a load would have raised)."""
    import json
    out = _run_bench(["--gpus", "2", "--config", "c3"], {"CBO_BENCH_DRY_RANKS": "1", "CBO_HIP_LIB": "/nonexistent/libcbo_hip.so"})
    assert out.returncode == 0, out.stderr[-2000:]
    r0 = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    r1 = [json.loads(ln) for ln in out.stderr.splitlines() if ln.startswith("{")]
    assert len(r0) == 1 and len(r1) == 1                          # rank 0 alone owns stdout
    a, b = r0[0], r1[0]
    assert (a["dry_rank"], a["local_rank"], a["world"]) == (0, 0, 2) and (b["dry_rank"], b["local_rank"], b["world"]) == (1, 1, 2)
    assert a["master"] == b["master"] and a["master"][0] == "127.0.0.1" and int(a["master"][1]) > 0
    assert a["launcher_pid"] == b["launcher_pid"]                  # the rendezvous file's key (sharding._id_path)
    assert a["shard"] == [0, 32768] and b["shard"] == [32768, 65536] and a["candidates_total"] == 65536
    assert a["n_obs"] == 8192 and a["scaling"] == "strong" and not a["lib_loaded"] and not b["lib_loaded"]
    # four ranks of the 8-GPU config: every rank keeps the 1/8 shard it has on 8 GPUs
    out = _run_bench(["--gpus", "4", "--config", "c4"], {"CBO_BENCH_DRY_RANKS": "1", "CBO_HIP_LIB": "/nonexistent/libcbo_hip.so"})
    assert out.returncode == 0, out.stderr[-2000:]
    shards = sorted(json.loads(ln)["shard"] for ln in (out.stdout + out.stderr).splitlines() if ln.startswith("{"))
    assert shards == [[32768 * r, 32768 * (r + 1)] for r in range(4)]
    # the headline config over four ranks carries both figures: the weak-scaling shards (one 16384-candidate grid per GPU,
    # `value`) and the strong-scaling shards of the config's ONE grid (the `strong` object of the line)
    out = _run_bench(["--gpus", "4"], {"CBO_BENCH_DRY_RANKS": "1", "CBO_HIP_LIB": "/nonexistent/libcbo_hip.so"})
    assert out.returncode == 0, out.stderr[-2000:]
    ranks = sorted((json.loads(ln) for ln in (out.stdout + out.stderr).splitlines() if ln.startswith("{")), key=lambda r: r["dry_rank"])
    assert [r["shard"] for r in ranks] == [[16384 * r, 16384 * (r + 1)] for r in range(4)]
    assert all(r["scaling"] == "weak" and r["candidates_total"] == 65536 for r in ranks)
    assert [r["strong"]["shard"] for r in ranks] == [[4096 * r, 4096 * (r + 1)] for r in range(4)]
    assert all(r["strong"]["candidates_total"] == 16384 for r in ranks)
    # one rank: nothing to cut
    out = _run_bench(["--gpus", "1"], {"CBO_BENCH_DRY_RANKS": "1", "CBO_BENCH_SELF_LAUNCH": "1", "CBO_HIP_LIB": "/nonexistent/libcbo_hip.so"})
    assert out.returncode == 0 and json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["strong"] is None


def test_bench_launcher_reports_the_worst_exit_code_of_its_ranks():
    """A rank that cannot load the library fails loudly (no CPU fallback) and the launcher's exit code says so."""
    out = _run_bench(["--gpus", "2", "--config", "c3", "--steps", "1"], {"CBO_HIP_LIB": "/nonexistent/libcbo_hip.so"})
    assert out.returncode != 0
    assert "libcbo_hip.so not found" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_schedule_tuner_settles_near_the_best_candidate_on_a_simulated_device(tmp_path):
    """cbo_gp_fit_sweep chooses its schedule by timing the caller's own calls (csrc/schedule_tuner.h: host logic, no HIP).
    Against simulated devices -- step-time curves with the valley at a quarter of the pairs, at three eighths, at the
    empty pipeline, at the plain sequence (by a little, and by far: fits that retry with jitter), at everything pipelined; 1 % noise; a penalty on the call after a change --
    it settles within its call budget and within 1.6 % of the best candidate."""
    import subprocess
    exe = tmp_path / "schedule_tuner_sim"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "cbo_with_oop_amd", "csrc"),
                           os.path.join(ROOT, "tests", "support", "schedule_tuner_sim.cpp"), "-o", str(exe)])
    shapes = [(4096, 16384), (8192, 16384), (2048, 262144), (8192, 4096), (16384, 16384), (512, 16384), (4096, 4096)]
    for curve in range(6):
        for n_pad, m_pad in shapes:
            if curve == 2 and m_pad // 64 >= 224:
                continue                  # (everything pipelined only wins when the strips cannot fill the device)
            for seed in (1, 2, 3):
                out = subprocess.check_output([str(exe), str(n_pad), str(m_pad), "256", "224", str(curve), "0.01", str(seed)],
                                              text=True).split()
                calls, pairs, group, settled = (int(t) for t in out[:4])
                chosen, best = float(out[4]), float(out[5])
                assert settled == 1 and calls <= 97, (curve, n_pad, m_pad, seed, out)
                assert chosen <= best * 1.016, (curve, n_pad, m_pad, seed, out)
    # a fresh context's first calls run on a device whose clocks are still rising (a trace: the first split 12 % slower on
    # calls 2-3 than once warm, the candidates measured ten calls later 3 %): the first split's sample is renewed before it
    # may lose, and the tuner still settles on the valley -- where the first split of this shape sits
    for seed in (1, 2, 3, 4, 5):
        out = subprocess.check_output([str(exe), "2048", "16384", "256", "224", "0", "0.01", str(seed)], text=True,
                                      env=dict(os.environ, TUNER_SIM_WARM="0.2")).split()
        assert int(out[3]) == 1 and int(out[0]) <= 97 and float(out[4]) <= float(out[5]) * 1.016, out
    # a shape next to a settled one starts from that one's split (same fraction of the pairs), not from the plain sequence;
    # a shape far away starts from the analytic split (round 4: from two timing calls of the plain sequence)
    near = subprocess.check_output([str(exe), "4096", "16384", "256", "224", "0", "0.01", "1", "24576"], text=True).split()
    assert int(near[0]) == 4 and int(near[2]) == 16, near                     # 4 of 16 pairs, as settled at 16384 candidates
    far = subprocess.check_output([str(exe), "4096", "16384", "256", "224", "0", "0.01", "1", "262144"], text=True).split()
    assert int(far[0]) == 0, far              # 4096 strips in 17 rounds: not one pair's pipeline fits beside the factorisation
    # a caller that is never sampled (every call under the profiling timers) on a shape nothing is known about runs the
    # analytic split from its first call to its thirtieth -- not the plain sequence (round 4's tuner left such a caller with
    # fit-then-sweep for ever); with fewer strips than CUs that split is "everything pipelined"
    first, last = (l.split() for l in subprocess.check_output([str(exe), "unsampled", "4096", "16384", "256", "224"], text=True).splitlines())
    assert first == last and 1 <= int(first[0]) < int(first[2]) and int(first[3]) == 0, (first, last)   # some pairs of 16, still COLD
    first, last = (l.split() for l in subprocess.check_output([str(exe), "unsampled", "4096", "4096", "256", "224"], text=True).splitlines())
    assert first == last and int(first[0]) == int(first[2]) == 16, (first, last)                        # 64 strips: all 16 pairs
    # a settled shape whose fits begin to need another number of jitter retries is measured afresh after three such calls
    # (a pipelined split repeats its pipeline with every retry): SETTLED (6), still SETTLED after two, COLD (0) after the
    # third, and the next call starts over from the analytic split (the sequence is timed again right behind it)
    drift = subprocess.check_output([str(exe), "drift", "4096", "16384", "256", "224"], text=True).split()
    assert drift[:3] == ["6", "6", "0"] and int(drift[3]) >= 1, drift
    # the per-context table of shapes is bounded
    assert int(subprocess.check_output([str(exe), "bounded"], text=True)) == 64


def test_logexp_transform_and_the_optimizer_call_are_paramz_s():
    """``model.optimize()`` (src/CBO.py:173, src/utils_functions/utils.py:44) is host logic over the device's likelihood and
    gradients: paramz's Logexp parametrisation (closed forms; the product's and the oracle's restatements agree) and the
    optimiser call (``fmin_l_bfgs_b(f_fp, finv(theta0), maxfun=max_iters, maxiter=max_iters)``, model left at x_opt).  The
    method is driven here with the ORACLE's likelihood in the device's place (no GPU): its trajectory must then be the
    oracle optimiser's, evaluation for evaluation."""
    from types import SimpleNamespace
    import importlib
    F = importlib.import_module("cbo_with_oop_amd.GaussianProcessFactory")     # (the package re-exports the class under this name)
    x = np.array([-800.0, -720.0, -30.0, -1.0, 0.0, 1.0, 35.0, 36.0, 36.5, 800.0])
    th = F.logexp_f(x)
    assert np.array_equal(th, O.logexp_f(x)) and np.all(th > 0)
    assert np.array_equal(th[x > 36], x[x > 36])                                 # identity above the limit
    mid = (x >= -30) & (x <= 36)
    assert np.allclose(th[mid], np.log1p(np.exp(x[mid])), rtol=1e-15)
    assert np.allclose(F.logexp_finv(th[mid]), x[mid], rtol=0, atol=1e-9)         # round trip
    assert np.array_equal(F.logexp_finv(th), O.logexp_finv(th))
    assert np.array_equal(F.logexp_gradfactor(th), O.logexp_gradfactor(th))
    xs = np.linspace(-20, 30, 101)
    fd = (F.logexp_f(xs + 1e-6) - F.logexp_f(xs - 1e-6)) / 2e-6                   # d theta / d x = 1 - exp(-theta)
    assert np.allclose(F.logexp_gradfactor(F.logexp_f(xs)), fd, rtol=1e-6, atol=1e-12)
    assert F.logexp_finv(np.array([1.0]))[0] == np.log(np.expm1(1.0))           # the reference's initial parameters

    rng = np.random.default_rng(5)
    X = rng.uniform(-4, 4, (40, 2))
    y = np.sin(1.3 * X[:, :1]) * np.cos(0.7 * X[:, 1:]) + 0.05 * rng.standard_normal((40, 1))

    def stand_in(fix_noise, ls0, noise0):
        m = SimpleNamespace(variance=1.0, lengthscale=np.atleast_1d(np.asarray(ls0, dtype=np.float64)), noise_var=noise0,
                            fix_noise=fix_noise, small=True, _last_lml=None, calls=0)
        def set_hyperparameters(v, ls, nz, fit=True):
            m.variance, m.lengthscale, m.noise_var = float(v), np.atleast_1d(np.asarray(ls, dtype=np.float64)), float(nz)
        def log_likelihood_gradients():
            m.calls += 1
            ls = m.lengthscale if m.lengthscale.size > 1 else float(m.lengthscale[0])
            post = O.fit(X, y, None, None, m.variance, ls, m.noise_var)
            m._last_lml = O.log_marginal_likelihood(post)
            dv, dls, dn = O.log_marginal_likelihood_gradients(post)
            return dv, np.atleast_1d(dls), dn
        m.set_hyperparameters, m.log_likelihood_gradients = set_hyperparameters, log_likelihood_gradients
        m._objective = lambda x, transform="log": F.HipGaussianProcess._objective(m, x, transform)
        return m

    for transform in ("logexp", "log"):
        for fix_noise, ls0, noise0 in ((False, 1.0, 1e-10), (True, np.ones(2), 1e-2)):
            m = stand_in(fix_noise, ls0, noise0)
            res = F.HipGaussianProcess.optimize(m, transform=transform)
            info = {}
            v, ls, nz, lml = O.optimize_hyperparameters(X, y, variance=1.0, lengthscale=ls0, noise_var=noise0,
                                                        fix_noise=fix_noise, transform=transform, info=info)
            assert res.nfev == info["funcalls"] and res.nit == info["nit"] and res.success
            assert m.calls == res.nfev + 1                                        # + opt_lbfgsb's f_fp(x_opt)
            assert (m.variance, m.noise_var) == (v, nz) and np.array_equal(m.lengthscale, ls)     # the model sits at x_opt
            assert -res.fun == lml and res.transform == transform
    # the two parametrisations reach the same stationary point by different paths
    a, b = stand_in(False, 1.0, 1e-10), stand_in(False, 1.0, 1e-10)
    F.HipGaussianProcess.optimize(a), F.HipGaussianProcess.optimize(b, transform="log")
    assert np.isclose(a.variance, b.variance, rtol=1e-3) and np.allclose(a.lengthscale, b.lengthscale, rtol=1e-3)
    with pytest.raises(ValueError):
        F.HipGaussianProcess.optimize(a, transform="softplus")
