#!/usr/bin/env python3
"""bench.py -- candidate-intervention acquisitions/sec of the MI355X hot path.  No PyTorch in this process.

One "step" = one pass of the whole hot path of BASELINE.json's north_star over one batch of synthetic
input, exactly what CBO.intervene() triggers each trial for the set it intervened on
(/root/reference/src/CBO.py:152-164):
    fit   : K(X,X) assembly -> jittered Cholesky with the forward solve z = L^-1 (y - m) carried along
            (GPy's alpha = L^-T z is not needed by the fp64 sweep and is materialised on demand)
    sweep : K(X,X*) -> V = L^-1 K* (variance, mean) -> EI / cost -> arg-max over the rank's candidates
    pick  : arg-max exchange across ranks (RCCL all-gather of 16 B per rank, formed inside libcbo_hip.so)
The timed steps make ONE device call for fit + sweep (cbo_gp_fit_sweep): in fp64 the sweep's substitution advances
pair of panels by pair on two extra streams underneath the factorisation's chain of short kernels.
--sequential times the same work as two calls (cbo_gp_fit, then cbo_acq_sweep), nothing overlapped.
Inputs (X, y, candidate grid) are resident in HBM before the timed region starts; the only host
traffic inside it is the jitchol status word and the 16-byte winner.

After the timed region a short instrumented pass (not part of `value`) runs the two-call sequence with
per-phase hipEvent timers on the library's own stream: it prices the phases and the dominant kernel on its own.

Workloads (--config; every BASELINE.json config has a line):
  c1 = configs[0]: toy_graph, 50 observations, 200-candidate sweep of each of its 2 exploration sets: one
       cbo_acq_sweep_sets call factors and sweeps every set in one launch (latency-bound; replicas over ranks).
  c2 = configs[1] (default, the config `metric` is quoted on): toy_graph box, d=3, 4096 observations, 16384-candidate
       regular grid (32x32x16), fp64.  Default scaling over ranks: weak (a 16384-candidate grid per GPU, as the metric's
       "16k grid at 1/2/4/8" has been measured since round 1); --scaling strong cuts the one grid into shards.
  c3 = configs[2]: complete_graph (B, D, E) ranges, 8192 observations, the fixed 65536-candidate grid (64x32x32) cut into
       one contiguous shard per rank (strong scaling; RCCL arg-max).
  c4 = configs[3]: simplified_coral_graph (N, O, T) ranges, 16384 observations, the fixed 262144-candidate grid (64^3)
       cut into one shard per rank, fp64.  The config is an 8-GPU one: with fewer than 8 ranks each rank still takes a
       1/8 shard (32768 candidates) unless --full-grid is given; `config.workload` says which.
  c5 = configs[4]: coral_graph (N, O, T) ranges, as c4 in fp32 (fp64 fit, fp32 sweep on the f32 MFMA).  --dtype f32
       is the same thing.

usage: python bench.py [--config c1..c5] [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--full-grid]
       (N > 1: either under python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... -- the launcher
        only provides RANK / WORLD_SIZE / LOCAL_RANK -- or as a plain `python bench.py --gpus N`, which then starts its
        own N rank processes as children (self_launch); the communicator is RCCL through the C-ABI either way)
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6                           # MI355X fp64 matrix peak (AMD datasheet; = vector peak)
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X f32-input MFMA peak (MI355X_MICROARCH.md)
# BASELINE.json configs[1..4].  Boxes: toy ranges X, Z (+ a third axis, graphs.ToyGraph); the reference's interventional
# ranges for the others (/root/reference/src/graphs/impl/CompleteGraph.py:106-112, SimplifiedCoralGraph.py:185-192,
# CoralGraph.py:177-184).  `nominal_gpus`: the GPU count the config is written for -- with fewer ranks each rank still
# takes 1/nominal of the fixed grid unless --full-grid.
CONFIGS = {
    "c2": dict(n_obs=4096, grid=(32, 32, 16), box=[(-5.0, 5.0), (-5.0, 20.0), (-5.0, 5.0)], dtype="f64", scaling="weak",
               nominal_gpus=1, name="toy_graph box d=3, 4096 obs, 16384-candidate regular grid (BASELINE.json configs[1])",
               cpu_sample=4096),
    "c3": dict(n_obs=8192, grid=(64, 32, 32), box=[(-5.0, 4.0), (-5.0, 5.0), (-6.0, 3.0)], dtype="f64", scaling="strong",
               nominal_gpus=1, name="complete_graph (B, D, E) ranges d=3, 8192 obs, 65536-candidate regular grid 64x32x32 "
                                    "(BASELINE.json configs[2])", cpu_sample=1024),
    "c4": dict(n_obs=16384, grid=(64, 64, 64), box=[(-2.0, 5.0), (3.0, 4.0), (2300.0, 2400.0)], dtype="f64",
               scaling="strong", nominal_gpus=8,
               name="simplified_coral_graph (N, O, T) ranges d=3, 16384 obs, 262144-candidate regular grid 64^3 "
                    "(BASELINE.json configs[3])", cpu_sample=128),
    "c5": dict(n_obs=16384, grid=(64, 64, 64), box=[(-2.0, 5.0), (2.0, 4.0), (2450.0, 2500.0)], dtype="f32",
               scaling="strong", nominal_gpus=8,
               name="coral_graph (N, O, T) ranges d=3, 16384 obs, 262144-candidate regular grid 64^3, fp32 sweep "
                    "(BASELINE.json configs[4])", cpu_sample=128),
}


def make_problem(cfg, world, scaling, full_grid):
    """(X, y, candidates of the whole job, grid shape, note).  Weak scaling stacks one grid per rank along the last
    axis; strong scaling keeps the config's grid and the caller cuts it into contiguous shards."""
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    box, n = cfg["box"], cfg["n_obs"]
    lo, hi = np.array([b[0] for b in box]), np.array([b[1] for b in box])
    X = np.random.default_rng(0).uniform(lo, hi, (n, 3))
    u = (X - lo) / (hi - lo)
    if cfg is CONFIGS["c2"]:
        y = (np.cos(np.exp(-X[:, 0] / 3)) - np.exp(-X[:, 1] / 20) + 0.3 * np.sin(X[:, 2])
             + 0.1 * np.random.default_rng(1).standard_normal(n))[:, None]
    else:
        y = (np.sin(3 * u[:, 0]) + np.cos(2 * u[:, 2]) * u[:, 1]
             + 0.05 * np.random.default_rng(1).standard_normal(n))[:, None]
    g = cfg["grid"]
    if scaling == "weak":
        grid = (g[0], g[1], g[2] * world)
        return X, y, meshgrid_candidates(box, grid), grid, f"one {g[0]}x{g[1]}x{g[2]} grid per GPU"
    Xs = meshgrid_candidates(box, g)
    nominal = cfg["nominal_gpus"]
    if world < nominal and not full_grid:
        share = Xs.shape[0] // nominal * world
        return X, y, Xs[:share], g, (f"the first {world}/{nominal} of the fixed {g[0]}x{g[1]}x{g[2]} grid: each rank takes the "
                                     f"1/{nominal} shard it has in the {nominal}-GPU configuration")
    return X, y, Xs, g, f"the fixed {g[0]}x{g[1]}x{g[2]} grid cut into {world} contiguous shard(s)"


def strong_shard(cfg, world, rank):
    """c2 over ranks, strong scaling: the config's ONE grid (16384 candidates) cut into `world` contiguous shards -- what
    the north_star's "the candidate-intervention grid shards across the GPUs" says (src/CBO.py:237-260: one loop over the
    candidates of a set).  Returns (candidates of the whole grid, begin, end of this rank's shard)."""
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    from cbo_with_oop_amd.sharding import shard_bounds
    Xs = meshgrid_candidates(cfg["box"], cfg["grid"])
    begin, end = shard_bounds(Xs.shape[0], world, rank)
    return Xs, int(begin), int(end)


def cpu_baseline(X, y, Xs, y_best, cost, sample):
    """numpy/scipy restatement of the GPy/emukit path (oracle/gp_oracle.py) timed on this box's host
    cores: the full fit, and the sweep on the first `sample` candidates scaled to the full grid."""
    from oracle import gp_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    t0 = time.perf_counter()
    post = O.fit(X, y)
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.acquisition_sweep(post, Xs[:sample], y_best, cost=cost)
    t_sweep = (time.perf_counter() - t0) * (Xs.shape[0] / sample)
    return {"value": Xs.shape[0] / (t_fit + t_sweep), "unit": "acquisitions/s", "cores": int(threads),
            "kind": "port",
            "sample": f"full fit N={X.shape[0]} ({t_fit:.2f} s) + sweep of the first {sample} of {Xs.shape[0]} "
                      f"candidates scaled x{Xs.shape[0] / sample:.0f} ({t_sweep:.2f} s); numpy/scipy restatement "
                      f"of the GPy/emukit path (fp64), not GPy itself" +
                      ("; a triangular solve with only this many right-hand sides runs below BLAS-3 speed, so this "
                       "baseline is understated (a stated baseline, never the target)" if sample < 512 else "")}


def kernel_sources_sha():
    """sha256 over the kernel and C-ABI sources: ties a committed PMC measurement to the code it was taken on."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "cbo_with_oop_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(section):
    """(HBM bytes, provenance) from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by
    scripts/pmc_to_json.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command).  The file records the
    hash of the kernel sources it was measured on; when that differs from the sources in this tree the number is
    stale and `traffic` is reported as null.  gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled
    (MI355X_MICROARCH.md HBM)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if section is None:
        return None, "no PMC pass was taken at this config's shape (profiles/pmc_traffic.json covers c2 and the c5 shard)"
    try:
        doc = json.load(open(path))
        d = doc[section]
        measured_on, now = doc.get("kernel_sources_sha"), kernel_sources_sha()
        if measured_on != now:
            return None, f"profiles/pmc_traffic.json was measured on kernel sources {measured_on}, this tree is {now}: stale"
        return d["fetch_size_kb"] * 1024 * 2 + d["write_size_kb"] * 1024, f"profiles/pmc_traffic.json @ {measured_on}"
    except Exception as e:  # noqa: BLE001
        return None, f"no PMC measurement for '{section}' ({type(e).__name__})"


def bench_small_sets(args):
    """--config c1 = BASELINE.json configs[0]: toy_graph, 50 observations and a 200-candidate sweep per exploration set
    (2 sets).  Step = what CBO.intervene() does per trial at that size (/root/reference/src/CBO.py:152-164): the model of
    the set intervened on is rebuilt, every set is factored and swept in one launch, the set is picked -- ONE library call
    (cbo_trial_step through CBOAcquisitionPath.trial_step).  Every rank runs the same pass (replicas: two 200-candidate
    sets are not worth sharding)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from cbo_with_oop_amd import CBOAcquisitionPath, GaussianProcessType, _lib
    from cbo_with_oop_amd.graphs import ToyGraph, meshgrid_candidates
    from cbo_with_oop_amd.sharding import Communicator
    ctx = _lib.Context.get(local_rank % max(1, _lib.device_count()))
    comm = quiet_communicator(Communicator, ctx)
    steps = args.steps or 2000
    rng = np.random.default_rng(0)
    es = ToyGraph.get_exploration_set("MIS")
    xs = [rng.uniform(-5, 5, (50, 1)), rng.uniform(-5, 20, (50, 1))]
    ys = [ToyGraph.target_do_x(xs[0]), ToyGraph.target_do_z(xs[1])]
    path = CBOAcquisitionPath(GaussianProcessType.NON_CAUSAL_GP, es, ToyGraph.get_cost_structure(1), "min", xs, ys,
                              [ToyGraph.bounds(s) for s in es], grid_shapes=[[200], [200]], comm=None)
    path.update_all_gaussian_processes()
    best = min(float(ys[0].min()), float(ys[1].min()))
    path.last_intervention = 1

    def step():
        path.last_intervention = 1                  # (the set whose model is rebuilt: the same every step)
        _, vals, choice = path.trial_step(best)     # one cbo_trial_step call: rebuild + sweep of every set + pick
        return choice, vals

    def fence():
        if comm is not None:
            comm.barrier()
        ctx.synchronize()

    for _ in range(max(5, args.warmup)):
        choice, vals = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        choice, vals = step()
    fence()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max(elapsed)
    if rank == 0:
        per = elapsed / steps
        n_sets, m, n = 2, 200, 50
        # algorithmic work of a pass: per set, factorisation n^3/3 + substitution n^2 per candidate (fp64 flops)
        flops = n_sets * (n ** 3 / 3.0 + float(n) ** 2 * m)
        out = {"metric": "candidate-intervention acquisitions/sec (toy_graph, 50 obs, 200 candidates x 2 sets)",
               "value": world * n_sets * m / per, "unit": "acquisitions/s", "n_gpus": world, "steps": steps,
               "warmup": max(5, args.warmup), "ms_per_step": per * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "toy_graph, 50 obs, 200-candidate sweep of each of the 2 exploration sets "
                                      "(BASELINE.json configs[0]); step = ONE cbo_trial_step call: rebuild the intervened "
                                      "set's model + every set factored and swept in one launch + pick",
                          "config": "c1", "n_obs": n, "candidates_total": n_sets * m, "sets": n_sets,
                          "rccl_ranks": comm.size()[0] if comm is not None else 0,
                          "parallelism": f"replicas x{world} (every rank runs the whole pass)"},
               "winner": {"set": int(choice[1]), "acq": float(vals[choice[1]][0, 0])},
               "kernel_sources_sha": kernel_sources_sha(),
               "roofline": {"kernel": "small_sets_kernel (K, factorisation, K*, substitution, EI, arg-max of a set in one "
                                      "workgroup per 64 candidates)", "bound": "mfma",
                            "achieved": flops / per / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": flops / per / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                            "note": "latency-bound by construction: 8 workgroups, 50 dependent pivots per set; the "
                                    "figure of merit is ms_per_step (host call included)",
                            "algorithmic_flops_per_step": flops}}
        if args.cpu_sample is None or args.cpu_sample > 0:
            from oracle import gp_oracle as O
            grids = [meshgrid_candidates(ToyGraph.bounds(s), [200]) for s in es]
            posts = [O.fit(xs[s], ys[s]) for s in range(2)]

            def cpu_pass():
                posts[1] = O.fit(xs[1], ys[1])
                v = [O.acquisition_sweep(posts[s], grids[s], best, cost=1.0)[1] for s in range(2)]
                return O.select_next_intervention([np.array([[a]]) for a in v])
            for _ in range(5):
                cpu_pass()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                cpu_choice = cpu_pass()
            dt = (time.perf_counter() - t0) / reps
            out["cpu_baseline"] = {"value": n_sets * m / dt, "unit": "acquisitions/s", "cores": 1, "kind": "port",
                                   "sample": f"the whole pass, {reps} repetitions ({dt * 1e3:.3f} ms each): refit of the "
                                             f"intervened set + sweep of both sets + pick; numpy/scipy restatement",
                                   "same_choice": bool(int(cpu_choice) == int(choice[1]))}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()
        comm.close()


def quiet_communicator(Communicator, ctx):
    """Communicator.from_env with file descriptor 1 pointed at stderr meanwhile: RCCL greets with a version banner on
    stdout when a communicator is formed, and stdout carries this program's one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        return Communicator.from_env(ctx)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def self_launch(n_ranks, argv=None, timeout_s=None):
    """`python bench.py --gpus N` with no launcher around it: start N rank processes of this same command as CHILDREN
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, what torch.distributed.run would
    set), hand rank 0's standard output through (its one JSON line), send the other ranks' output to stderr, and return the
    worst exit code.  The launcher itself never imports the library or initialises the GPU (a process that has must not
    exec or fork workers on this pool); the ranks form their RCCL communicator among themselves (sharding.Communicator:
    the 128-byte id travels through a private file keyed by this launcher's pid and MASTER_PORT)."""
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                LOCAL_WORLD_SIZE=str(n_ranks))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between processes on this driver
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr, stderr=None))
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    worst, failed_at = 0, None
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0:
                    worst = worst or rc
                    failed_at = failed_at or time.monotonic()
            if live:
                # a rank that died leaves the others in a collective: give them a minute, then stop exactly them
                if (failed_at and time.monotonic() - failed_at > 60) or (deadline and time.monotonic() > deadline):
                    for p in live:
                        p.terminate()
                    for p in live:
                        try:
                            p.wait(10)
                        except subprocess.TimeoutExpired:
                            p.kill()
                    worst = worst or 124
                    break
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def dry_rank_report(cfg, world, rank, local_rank, scaling, full_grid):
    """CBO_BENCH_DRY_RANKS=1 (tests/test_host_logic.py): a rank says what it was launched as and which shard of which
    problem it would sweep, without loading libcbo_hip.so or touching a GPU."""
    from cbo_with_oop_amd.sharding import shard_bounds
    X, y, Xs, grid, note = make_problem(cfg, world, scaling, full_grid)
    begin, end = shard_bounds(Xs.shape[0], world, rank)
    strong = None
    if cfg is CONFIGS["c2"] and scaling == "weak" and world > 1:      # the line carries a strong-scaling object as well
        Xg, sb, se = strong_shard(cfg, world, rank)
        strong = {"shard": [sb, se], "candidates_total": int(Xg.shape[0])}
    print(json.dumps({"dry_rank": rank, "world": world, "local_rank": local_rank, "shard": [int(begin), int(end)],
                      "strong": strong,
                      "candidates_total": int(Xs.shape[0]), "n_obs": int(X.shape[0]), "scaling": scaling,
                      "master": [os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")],
                      "launcher_pid": os.getppid(), "lib_loaded": "cbo_with_oop_amd._lib" in sys.modules and
                      sys.modules["cbo_with_oop_amd._lib"]._lib is not None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20 for c2, fewer for the larger configs)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default=None,
                    help="BASELINE.json config (default c2, the one the metric is quoted on)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default=None, help="f32 = --config c5")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="over ranks: weak = the config's grid per GPU, strong = the config's one grid cut into shards "
                         "(default: weak for c2, strong for c3-c5)")
    ap.add_argument("--full-grid", action="store_true",
                    help="c4/c5 with fewer than 8 ranks: sweep the whole 262144-candidate grid instead of 1/8 per rank")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="candidates in the CPU-baseline sample (0 = skip; default per config: about 10-30 s of sweep next to\n"
                         "the full CPU fit)")
    ap.add_argument("--post-steps", type=int, default=3,
                    help="instrumented two-call steps after the timed region (phase timers, isolated kernel; 0 = skip)")
    ap.add_argument("--ladder-over-ranks", action="store_true",
                    help="walk jitchol's ladder with one level per rank (sharding.fit_over_ranks) instead of every rank "
                         "walking all of it: the step is that fit, then cbo_acq_sweep.  Off by default: the factor "
                         "hand-over (ncclSend / ncclRecv) has only ever run on one rank")
    ap.add_argument("--sequential", action="store_true",
                    help="refit, then sweep (two calls) instead of the one cbo_gp_fit_sweep call")
    args = ap.parse_args()
    if args.config is None:
        args.config = "c5" if args.dtype == "f32" else "c2"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("CBO_BENCH_SELF_LAUNCH") == "1"):
        # a plain `python bench.py --gpus N`: this process becomes the launcher (it has not loaded libcbo_hip.so and
        # never touches the GPU) and starts one fresh rank process per GPU (CBO_BENCH_SELF_LAUNCH=1: also for N = 1, so
        # that a one-GPU box can exercise the launcher and a one-rank RCCL communicator)
        sys.exit(self_launch(args.gpus))
    if args.config == "c1":
        return bench_small_sets(args)
    cfg = CONFIGS[args.config]
    if args.dtype is not None and args.dtype != cfg["dtype"]:
        sys.exit(f"--config {args.config} is an {cfg['dtype']} workload")
    args.dtype = cfg["dtype"]
    f32 = args.dtype == "f32"
    scaling = args.scaling or cfg["scaling"]
    if args.steps is None:
        args.steps = 20 if args.config == "c2" else (8 if cfg["n_obs"] <= 8192 else 4)
    if args.cpu_sample is None:
        args.cpu_sample = cfg["cpu_sample"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                                      # (a plain run with --gpus N > 1 became N ranks in main())
    if os.environ.get("CBO_BENCH_DRY_RANKS"):
        return dry_rank_report(cfg, world, rank, local_rank, scaling, args.full_grid)

    from cbo_with_oop_amd import CandidateGrid, CausalExpectedImprovement, _lib
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    from cbo_with_oop_amd.sharding import Communicator, exchange_argmax, fit_over_ranks, shard_bounds

    ctx = _lib.Context.get(local_rank % max(1, _lib.device_count()))
    lib = _lib.load()
    # under a one-process-per-GPU launcher the communicator is always formed (also for one rank, so that the RCCL
    # exchange is exercised on a one-GPU box); a plain `python bench.py` has none
    comm = quiet_communicator(Communicator, ctx)
    X, y, Xs, grid, grid_note = make_problem(cfg, world, scaling, args.full_grid)
    n_obs = X.shape[0]
    per_gpu = cfg["grid"]
    y_best, cost = float(y.min()), 3.0                     # incumbent = best observation; type_cost 1 -> |set| = 3
    begin, end = shard_bounds(Xs.shape[0], world, rank)

    model = HipGaussianProcess(X, y, context=ctx, fit=False, dtype=args.dtype)   # uploads X, y; the first warm-up step fits
    cands = CandidateGrid(Xs[begin:end], model, index_offset=begin, context=ctx)
    bv, bi = ctypes.c_double(), ctypes.c_int64()

    def step():
        if args.ladder_over_ranks:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)       # ("Added jitter of ...": the ladder at work)
                fit_over_ranks(model, comm)
            _lib.check(lib.cbo_acq_sweep(model._handle, cands._handle, y_best, 0, 0.0, cost, None, None, None,
                                         ctypes.byref(bv), ctypes.byref(bi)))
        elif args.sequential:
            _lib.check(lib.cbo_gp_fit(model._handle, None, None))
            _lib.check(lib.cbo_acq_sweep(model._handle, cands._handle, y_best, 0, 0.0, cost, None, None, None,
                                         ctypes.byref(bv), ctypes.byref(bi)))
        else:
            _lib.check(lib.cbo_gp_fit_sweep(model._handle, cands._handle, y_best, 0, 0.0, cost, None, None, None,
                                            ctypes.byref(bv), ctypes.byref(bi), None, None))
        return exchange_argmax(bv.value, bi.value, comm)

    def fence():
        if comm is not None:
            comm.barrier()
        ctx.synchronize()                      # hipDeviceSynchronize

    def step_two_calls():
        _lib.check(lib.cbo_gp_fit(model._handle, None, None))
        _lib.check(lib.cbo_acq_sweep(model._handle, cands._handle, y_best, 0, 0.0, cost, None, None, None,
                                     ctypes.byref(bv), ctypes.byref(bi)))
        return bv.value, bi.value

    # cbo_gp_fit_sweep settles its schedule for this shape by timing its first calls (the plain sequence, then neighbouring
    # splits: cbo_api.hip, schedule_choose): those calls come before the warm-up, untimed and without the exchange (every
    # rank settles on its own)
    settling_calls = 0
    if not args.sequential and not args.ladder_over_ranks:
        _lib.check(lib.cbo_gp_fit_sweep(model._handle, cands._handle, y_best, 0, 0.0, cost, None, None, None,
                                        ctypes.byref(bv), ctypes.byref(bi), None, None))
        while ctx.schedule_report()[0] > 0 and settling_calls < 80:
            _lib.check(lib.cbo_gp_fit_sweep(model._handle, cands._handle, y_best, 0, 0.0, cost, None, None, None,
                                            ctypes.byref(bv), ctypes.byref(bi), None, None))
            settling_calls += 1
    for _ in range(max(1, args.warmup)):       # at least one: the model is created unfitted
        winner = step()
    fence()
    ctx.region_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        winner = step()
    region_ms = ctx.region_end()               # hipEvents on the ctx stream around the K steps
    fence()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max(elapsed)            # the slowest rank's time

    # c2 over several ranks: `value` above is the weak-scaling figure (one 16384-candidate grid per GPU, as measured since
    # round 1).  The same step on the config's ONE grid cut into `world` shards is timed here as well, by the same rules
    # (schedule settled first, warm-up, barrier + device synchronisation on both sides, the slowest rank's time): with the
    # posterior replicated, every rank still factors all 4096 rows -- the Amdahl term of this design.
    strong = None
    # (CBO_BENCH_STRONG_AT_ONE=1 takes this branch with one rank as well, so that a one-GPU box executes it: the test suite)
    several = world > 1 or os.environ.get("CBO_BENCH_STRONG_AT_ONE") == "1"
    if args.config == "c2" and scaling == "weak" and several and not args.sequential and not args.ladder_over_ranks:
        Xg, sb, se = strong_shard(cfg, world, rank)
        cands_s = CandidateGrid(Xg[sb:se], model, index_offset=sb, context=ctx)

        def strong_step():
            _lib.check(lib.cbo_gp_fit_sweep(model._handle, cands_s._handle, y_best, 0, 0.0, cost, None, None, None,
                                            ctypes.byref(bv), ctypes.byref(bi), None, None))
            return exchange_argmax(bv.value, bi.value, comm)
        strong_settling = 0
        strong_step()
        while ctx.schedule_report()[0] > 0 and strong_settling < 80:
            strong_step()
            strong_settling += 1
        for _ in range(max(1, args.warmup)):
            strong_winner = strong_step()
        fence()
        ts = time.perf_counter()
        for _ in range(args.steps):
            strong_winner = strong_step()
        fence()
        strong_elapsed = comm.max(time.perf_counter() - ts) if comm is not None else time.perf_counter() - ts
        strong = {"scaling": "strong", "candidates_total": int(Xg.shape[0]), "candidates_per_gpu": int(-(-Xg.shape[0] // world)),
                  "shard_of_rank_0": [sb, se], "ms_per_step": strong_elapsed / args.steps * 1e3,
                  "value": Xg.shape[0] / (strong_elapsed / args.steps), "unit": "acquisitions/s",
                  "winner": {"index": int(strong_winner[1]), "acq": float(strong_winner[0])},
                  "schedule_settling_calls": strong_settling,
                  "what": "the config's ONE 32x32x16 grid cut into contiguous shards, one per rank; the posterior (the "
                          "4096-row factorisation) is replicated on every rank, so the step cannot fall below the "
                          "factorisation's own time: DESIGN.md 6 has the modelled curve"}
        cands_s.close()

    # instrumented pass, outside the timed region: the same work as two calls, per-phase event timers on
    timers, post_winner = None, None
    if args.post_steps > 0:
        post_winner = step_two_calls()
        ctx.set_profiling(True)
        ctx.reset_timers()
        for _ in range(args.post_steps):
            post_winner = step_two_calls()
        timers = ctx.timers()
        ctx.set_profiling(False)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_cands = Xs.shape[0]
        n_pad = -(-n_obs // 128) * 128
        n_sweep = -(-n_pad // 256) * 256 if f32 else n_pad       # the fp32 layout pads the rows to 256
        m_rank = -(-(end - begin) // 64) * 64
        peak = FP32_MFMA_PEAK_TFLOPS if f32 else FP64_MFMA_PEAK_TFLOPS
        sweep_flops = float(n_sweep) ** 2 * m_rank               # substitution: n^2 flops per candidate column
        chol_flops = float(n_pad) ** 3 / 3.0
        metric = {"c2": "candidate-intervention acquisitions/sec (16k grid, d=3)",
                  "c3": "candidate-intervention acquisitions/sec (complete_graph, 8k obs, 64k grid, d=3)",
                  "c4": "candidate-intervention acquisitions/sec (simplified_coral_graph, 16k obs, 256k grid, d=3)",
                  "c5": "candidate-intervention acquisitions/sec (coral_graph fp32 path, 16k obs, 256k grid, d=3)"}[args.config]
        out = {
            "metric": metric,
            "value": total_cands / (elapsed / args.steps),
            "unit": "acquisitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
        }
        exchange = (f"RCCL all-gather of (val, idx) over {world} rank(s), inside libcbo_hip.so" if comm is not None
                    else "none (single process)")
        if args.ladder_over_ranks:
            step_mode = "jitchol's ladder one level per rank (fit_over_ranks), then cbo_acq_sweep"
        elif args.sequential:
            step_mode = "two calls, nothing overlapped"
        elif f32:
            step_mode = "cbo_gp_fit_sweep on an fp32 model: fp64 fit, one down-conversion of the factor, fp32 sweep"
        else:
            step_mode = ("cbo_gp_fit_sweep: right-looking sweep pairs under the factorisation (4 streams), left-looking "
                         "launch for the rest -- or the plain sequence; the split is the one the context measured "
                         "(config.schedule)")
        out["config"] = {
            "workload": f"{cfg['name']}; {grid_note}; step = " +
                        ("fp64 GP refit + fp32 EI/cost sweep (f32 MFMA) + argmax" if f32 else
                         "GP refit + EI/cost sweep + argmax"),
            "config": args.config, "n_obs": n_obs, "candidates_total": int(total_cands), "grid": list(grid),
            "candidates_per_gpu": int(-(-total_cands // world)),
            "step_mode": step_mode, "exchange": exchange,
            "rccl_ranks": comm.size()[0] if comm is not None else 0,     # what the communicator itself reports
            # the schedule cbo_gp_fit_sweep measured its way to before the warm-up (rank 0's; cbo_schedule_report)
            "schedule_settling_calls": settling_calls, "schedule": ctx.schedule_report()[1].strip(),
            "parallelism": f"candidate shards x{world}, replicated posterior"}
        out["winner"] = {"index": int(winner[1]), "acq": float(winner[0])}
        if strong is not None:
            out["strong"] = strong                    # (c2, several ranks: the one-grid figure beside the weak `value`)
        out["kernel_sources_sha"] = kernel_sources_sha()
        if f32:
            # the sweep kernel runs alone on the device (nothing overlaps on this dtype): its own launch duration is
            # its efficiency.  Dominant kernel = trsm_strip_f32_kernel, timed by hipEvents in the instrumented pass.
            if timers is not None and timers["n_trsm_launches"] > 0:
                launches = timers["n_trsm_launches"]
                trsm_ms = timers["ms_trsm"] / launches
                achieved = timers["trsm_flops"] / launches / (trsm_ms * 1e-3) / 1e12
                traffic, note = pmc_traffic("f32_strip_kernel" if total_cands // world == 32768 else None)
                out["roofline"] = {
                    "kernel": "trsm_strip_f32_kernel (V = L^-1 K* on v_mfma_f32_16x16x4_f32, fused sum V^2)",
                    "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "traffic_source": note, "per": "launch", "avg_launch_ms": trsm_ms,
                    "algorithmic_flops_per_launch": timers["trsm_flops"] / launches,
                    "whole_step_tflops_mixed": (sweep_flops + chol_flops) / (region_ms / args.steps * 1e-3) / 1e12}
            else:
                out["roofline"] = None
        else:
            # With the factorisation and the sweep co-scheduled a kernel's own launch duration measures its share
            # of the machine, not its efficiency: the roofline of the timed region is taken over the whole step
            # (all fp64-MFMA flops of a step / device time of a step, hipEvents around the K steps).
            step_flops = sweep_flops + chol_flops
            step_tflops = step_flops / (region_ms / args.steps * 1e-3) / 1e12
            traffic, note = pmc_traffic(("step_sequential" if args.sequential else "step") if args.config == "c2" else None)
            out["roofline"] = {
                "kernel": "whole step: trsm_update_kernel<16> (dominant) + trsm_strip_kernel<*,16> + trsm_pair_kernel + "
                          "syrk_kernel<64> + potrf_panel_fused_kernel (diagonal block + row panel), co-scheduled" if not args.sequential else
                          "whole step: trsm_pair_kernel<true> (dominant), then the factorisation's kernels",
                "bound": "mfma", "achieved": step_tflops, "peak": peak, "unit": "TFLOP/s",
                "frac": step_tflops / peak, "traffic": traffic, "traffic_source": note,
                "per": "step", "avg_step_ms_events": region_ms / args.steps,
                "algorithmic_flops_per_step": step_flops}
        if timers is not None and not f32:
            launches = max(1, timers["n_trsm_launches"])
            trsm_ms = timers["ms_trsm"] / launches
            achieved = timers["trsm_flops"] / launches / (trsm_ms * 1e-3) / 1e12 if trsm_ms > 0 else 0.0
            traffic, note = pmc_traffic("strip_kernel" if args.config == "c2" else None)
            out["roofline"]["isolated"] = {
                "kernel": "trsm_pair_kernel<true> (V = L^-1 K*, fused sum V^2 and V^T z; 256-row pair blocks, two waves per "
                          "SIMD): the sweep of every set that is not refitted, alone on the device (instrumented pass after "
                          "the timed region)",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": note,
                "avg_launch_ms": trsm_ms, "algorithmic_flops_per_launch": timers["trsm_flops"] / launches}
        if timers is not None:
            keys = ("ms_kxx", "ms_chol", "ms_alpha", "ms_kstar", "ms_trsm", "ms_acq", "ms_f32_convert")
            out["phases_ms_per_step_two_calls"] = {k[3:]: timers[k] / args.post_steps for k in keys}
            out["phases_ms_per_step_two_calls"]["chol_tflops"] = \
                chol_flops / (timers["ms_chol"] / args.post_steps * 1e-3) / 1e12
            # rank 0's own shard winner of the two-call pass; with one rank it is the winner of the timed steps
            out["two_calls_winner_matches"] = (int(post_winner[1]) == int(winner[1])) if world == 1 else None
            # secondary roofline (HBM-bound): K(X,X) assembly writes the upper 64x64 tiles of Ky once
            nt = n_pad // 64
            kxx_bytes = nt * (nt + 1) // 2 * 64 * 64 * 8
            kxx_ms = timers["ms_kxx"] / max(1, timers["n_fit"])
            out["roofline_kxx"] = {"kernel": f"kmat_tile_kernel<3> (K(X,X) + diag, upper tiles) + rhs at {n_obs} points",
                                   "bound": "hbm",
                                   "achieved": kxx_bytes / (kxx_ms * 1e-3) / 1e9 if kxx_ms > 0 else 0.0, "peak": 8000.0,
                                   "unit": "GB/s", "frac": (kxx_bytes / (kxx_ms * 1e-3) / 1e9 / 8000.0) if kxx_ms > 0 else 0.0,
                                   "avg_launch_ms": kxx_ms, "algorithmic_bytes_per_launch": kxx_bytes,
                                   "note": "fp64 exp per element: ALU-bound below the HBM roof (DESIGN.md 4)"}
        if world == 1 and args.post_steps > 0 and args.config == "c2":
            # the north-star's HBM-bound size: K(X,X) assembly for 16384 points (one fit of such a model)
            n16 = 16384
            lo_box, hi_box = np.array([b[0] for b in cfg["box"]]), np.array([b[1] for b in cfg["box"]])
            X16 = np.random.default_rng(2).uniform(lo_box, hi_box, (n16, 3))
            y16 = np.sin(X16).sum(1, keepdims=True)
            m16 = HipGaussianProcess(X16, y16, context=ctx, noise_var=1e-2, fit=False)
            _lib.check(lib.cbo_gp_fit(m16._handle, None, None))
            ctx.set_profiling(True)
            ctx.reset_timers()
            _lib.check(lib.cbo_gp_fit(m16._handle, None, None))
            t16 = ctx.timers()
            ctx.set_profiling(False)
            m16.close()
            nt16 = n16 // 64
            b16 = nt16 * (nt16 + 1) // 2 * 64 * 64 * 8
            out["roofline_kxx_16k"] = {"kernel": "kmat_tile_kernel<3> + rhs at 16384 points (BASELINE north_star size)",
                                       "bound": "hbm", "achieved": b16 / (t16["ms_kxx"] * 1e-3) / 1e9, "peak": 8000.0,
                                       "unit": "GB/s", "frac": b16 / (t16["ms_kxx"] * 1e-3) / 1e9 / 8000.0,
                                       "avg_launch_ms": t16["ms_kxx"], "algorithmic_bytes_per_launch": b16,
                                       "cholesky_ms": t16["ms_chol"],
                                       "cholesky_tflops": float(n16) ** 3 / 3.0 / (t16["ms_chol"] * 1e-3) / 1e12}
        if world == 1 and args.post_steps > 0 and args.config == "c2":
            # the north-star's other HBM-bound piece: the EI / cost / arg-max pass alone (3 * 8 B per
            # candidate -- q and mu in, the acquisition value out -- against 8 TB/s), on stored q, mu of 2^20 and 2^24
            # candidates (a re-sweep of an unchanged model skips the substitution: the epilogue is all that runs)
            out["roofline_ei"] = []
            Xe = np.random.default_rng(3).uniform(-5.0, 5.0, (64, 3))
            ye = np.sin(Xe).sum(1, keepdims=True)
            me = HipGaussianProcess(Xe, ye, context=ctx)
            for log2m in (20, 24):
                m_ei = 1 << log2m
                Ce = np.random.default_rng(4).uniform(-5.0, 5.0, (m_ei, 3))
                ge = CandidateGrid(Ce, me, context=ctx)
                acq_host = np.empty(m_ei)
                bve, bie = ctypes.c_double(), ctypes.c_int64()

                def ei_pass():
                    _lib.check(lib.cbo_acq_sweep(me._handle, ge._handle, float(ye.min()), 0, 0.0, 3.0, _lib.dptr(acq_host),
                                                 None, None, ctypes.byref(bve), ctypes.byref(bie)))
                ei_pass()                          # the substitution, once; q and mu stay with the candidates
                ei_pass()
                ctx.set_profiling(True)
                ctx.reset_timers()
                reps_ei = 10
                for _ in range(reps_ei):
                    ei_pass()
                te = ctx.timers()
                ctx.set_profiling(False)
                ms_ei = te["ms_acq"] / reps_ei
                out["roofline_ei"].append({
                    "kernel": "acq_kernel + argmax_final_kernel (variance, mean, EI / cost, arg-max from stored q, mu)",
                    "candidates": m_ei, "bound": "hbm", "achieved": 24.0 * m_ei / (ms_ei * 1e-3) / 1e9, "peak": 8000.0,
                    "unit": "GB/s", "frac": 24.0 * m_ei / (ms_ei * 1e-3) / 1e9 / 8000.0, "avg_pass_ms": ms_ei,
                    "algorithmic_bytes_per_pass": 24 * m_ei, "substitutions_in_the_timed_passes": int(te["n_trsm_launches"])})
                ge.close()
                del Ce, acq_host
            me.close()
        if world == 1 and args.post_steps > 0 and args.config == "c2":
            # what a CBO trial costs once data only grow by one observation (not part of `value`: the timed steps
            # refit from scratch): append one point to a 4000-point model, then sweep the same 16384-candidate grid
            n0 = 4000
            mA = HipGaussianProcess(X[:n0], y[:n0], context=ctx)
            gA = CandidateGrid(Xs[:per_gpu[0] * per_gpu[1] * per_gpu[2]], mA, context=ctx, keep_solution=True)
            eiA = CausalExpectedImprovement(y_best, "min", mA)
            eiA.sweep(gA, cost=cost)
            per_trial = []
            for i in range(n0, n0 + 24):
                t1 = time.perf_counter()
                ok = mA.append(X[i:i + 1], y[i:i + 1])
                eiA.sweep(gA, cost=cost)
                per_trial.append(time.perf_counter() - t1)
                assert ok
            out["append_trial_step"] = {"ms_per_trial": float(np.median(per_trial) * 1e3), "n_obs": n0 + 24,
                                        "candidates": len(gA),
                                        "what": "cbo_gp_append (one new column of the factor) + sweep that adds one row to "
                                                "the resident L^-1 K*; same results as a refit to rounding; excludes the "
                                                "per-trial hyper-parameter MLE of src/CBO.py:173"}
            gA.close()
            mA.close()
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(X, y, Xs, y_best, cost, min(args.cpu_sample, Xs.shape[0]))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    cands.close()
    model.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
