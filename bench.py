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

Workloads (weak scaling: fixed work per GPU, the posterior replicated, the grid cut into contiguous blocks):
  --dtype f64 (default) = BASELINE.json configs[1]: toy_graph box, d=3, 4096 observations, 16384-candidate regular grid
                (32x32x16) per GPU, fp64.
  --dtype f32 = BASELINE.json configs[4] at its per-GPU shard shape: coral_graph (N, O, T) ranges, 16384 observations,
                32768 candidates (32x32x32) per GPU; fp64 fit, fp32 sweep on the f32 MFMA.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f64|f32]
       (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...  -- the launcher only
        provides RANK / WORLD_SIZE / LOCAL_RANK; the communicator is RCCL through the C-ABI)
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_OBS = 4096
GRID_PER_GPU = (32, 32, 16)
BOX = [(-5.0, 5.0), (-5.0, 20.0), (-5.0, 5.0)]        # toy ranges X, Z (+ a third axis, see graphs.ToyGraph)
FP64_MFMA_PEAK_TFLOPS = 78.6                           # MI355X fp64 matrix peak (AMD datasheet; = vector peak)
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X f32-input MFMA peak (MI355X_MICROARCH.md)
# --dtype f32 = BASELINE.json configs[4] (coral_graph, fp32 path with MFMA) at its per-GPU shard shape: the (N, O, T)
# exploration set's ranges (/root/reference/src/graphs/impl/CoralGraph.py:177-184), 16384 observations, 256k
# candidates over 8 GPUs = a 32x32x32 grid per GPU
N_OBS_F32 = 16384
GRID_PER_GPU_F32 = (32, 32, 32)
BOX_F32 = [(-2.0, 5.0), (2.0, 4.0), (2450.0, 2500.0)]


def make_problem(world, dtype="f64"):
    from cbo_with_oop_amd.graphs import meshgrid_candidates
    if dtype == "f32":
        lo, hi = np.array([b[0] for b in BOX_F32]), np.array([b[1] for b in BOX_F32])
        X = np.random.default_rng(0).uniform(lo, hi, (N_OBS_F32, 3))
        u = (X - lo) / (hi - lo)
        y = (np.sin(3 * u[:, 0]) + np.cos(2 * u[:, 2]) * u[:, 1]
             + 0.05 * np.random.default_rng(1).standard_normal(N_OBS_F32))[:, None]
        grid = (GRID_PER_GPU_F32[0], GRID_PER_GPU_F32[1], GRID_PER_GPU_F32[2] * world)
        return X, y, meshgrid_candidates(BOX_F32, grid), grid
    lo, hi = np.array([b[0] for b in BOX]), np.array([b[1] for b in BOX])
    X = np.random.default_rng(0).uniform(lo, hi, (N_OBS, 3))
    y = (np.cos(np.exp(-X[:, 0] / 3)) - np.exp(-X[:, 1] / 20) + 0.3 * np.sin(X[:, 2])
         + 0.1 * np.random.default_rng(1).standard_normal(N_OBS))[:, None]
    grid = (GRID_PER_GPU[0], GRID_PER_GPU[1], GRID_PER_GPU[2] * world)
    return X, y, meshgrid_candidates(BOX, grid), grid


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
                      f"of the GPy/emukit path (fp64), not GPy itself"}


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
    try:
        doc = json.load(open(path))
        d = doc[section]
        measured_on, now = doc.get("kernel_sources_sha"), kernel_sources_sha()
        if measured_on != now:
            return None, f"profiles/pmc_traffic.json was measured on kernel sources {measured_on}, this tree is {now}: stale"
        return d["fetch_size_kb"] * 1024 * 2 + d["write_size_kb"] * 1024, f"profiles/pmc_traffic.json @ {measured_on}"
    except Exception as e:  # noqa: BLE001
        return None, f"no PMC measurement for '{section}' ({type(e).__name__})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20 for f64, 8 for f32)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--cpu-sample", type=int, default=None,
                    help="candidates in the CPU-baseline sample (0 = skip; default 4096 for f64, 64 for f32: ~30 s of sweep each\n"
                         "next to the full CPU fit)")
    ap.add_argument("--post-steps", type=int, default=3,
                    help="instrumented two-call steps after the timed region (phase timers, isolated kernel; 0 = skip)")
    ap.add_argument("--sequential", action="store_true",
                    help="refit, then sweep (two calls) instead of the one cbo_gp_fit_sweep call")
    args = ap.parse_args()
    f32 = args.dtype == "f32"
    if args.steps is None:
        args.steps = 8 if f32 else 20
    if args.cpu_sample is None:
        args.cpu_sample = 64 if f32 else 4096          # the 16384-point fit alone is ~25 s of CPU; keep the sample ~30 s

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    from cbo_with_oop_amd import CandidateGrid, CausalExpectedImprovement, _lib
    from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
    from cbo_with_oop_amd.sharding import Communicator, exchange_argmax, shard_bounds

    ctx = _lib.Context.get(local_rank % max(1, _lib.device_count()))
    lib = _lib.load()
    # under a one-process-per-GPU launcher the communicator is always formed (also for one rank, so that the RCCL
    # exchange is exercised on a one-GPU box); a plain `python bench.py` has none
    comm = Communicator.from_env(ctx)
    X, y, Xs, grid = make_problem(world, args.dtype)
    n_obs = X.shape[0]
    per_gpu = GRID_PER_GPU_F32 if f32 else GRID_PER_GPU
    y_best, cost = float(y.min()), 3.0                     # incumbent = best observation; type_cost 1 -> |set| = 3
    begin, end = shard_bounds(Xs.shape[0], world, rank)

    model = HipGaussianProcess(X, y, context=ctx, fit=False, dtype=args.dtype)   # uploads X, y; the first warm-up step fits
    cands = CandidateGrid(Xs[begin:end], model, index_offset=begin, context=ctx)
    bv, bi = ctypes.c_double(), ctypes.c_int64()

    def step():
        if args.sequential:
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
        out = {
            "metric": "candidate-intervention acquisitions/sec (16k grid, d=3)",
            "value": total_cands / (elapsed / args.steps),
            "unit": "acquisitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
        }
        exchange = (f"RCCL all-gather of (val, idx) over {world} rank(s), inside libcbo_hip.so" if comm is not None
                    else "none (single process)")
        if f32:
            out["metric"] = ("candidate-intervention acquisitions/sec (coral_graph fp32 path, 32k-candidate shard of the "
                             "256k grid, d=3)")
            out["config"] = {
                "workload": "coral_graph (N, O, T) ranges, d=3, 16384 obs, 32768-candidate regular grid per GPU = the "
                            "per-GPU shard of BASELINE.json configs[4] (256k candidates over 8 GPUs); step = fp64 GP refit "
                            "+ fp32 EI/cost sweep (f32 MFMA) + argmax",
                "n_obs": n_obs, "candidates_total": int(total_cands), "grid": list(grid),
                "candidates_per_gpu": int(total_cands // world),
                "step_mode": "two calls, nothing overlapped" if args.sequential else
                             "cbo_gp_fit_sweep on an fp32 model: fp64 fit, one down-conversion of the factor, fp32 sweep",
                "exchange": exchange, "parallelism": f"candidate shards x{world}, replicated posterior"}
        else:
            out["config"] = {
                "workload": "toy_graph box d=3, 4096 obs, 16384-candidate regular grid per GPU "
                            "(BASELINE.json configs[1]); step = GP refit + EI/cost sweep + argmax",
                "n_obs": n_obs, "candidates_total": int(total_cands), "grid": list(grid),
                "candidates_per_gpu": int(total_cands // world),
                "step_mode": "two calls, nothing overlapped" if args.sequential else
                             "cbo_gp_fit_sweep: right-looking sweep pairs under the factorisation (4 streams), "
                             "left-looking launch for the rest",
                "exchange": exchange, "parallelism": f"candidate shards x{world}, replicated posterior"}
        out["winner"] = {"index": int(winner[1]), "acq": float(winner[0])}
        out["kernel_sources_sha"] = kernel_sources_sha()
        if f32:
            # the sweep kernel runs alone on the device (nothing overlaps on this dtype): its own launch duration is
            # its efficiency.  Dominant kernel = trsm_strip_f32_kernel, timed by hipEvents in the instrumented pass.
            if timers is not None and timers["n_trsm_launches"] > 0:
                launches = timers["n_trsm_launches"]
                trsm_ms = timers["ms_trsm"] / launches
                achieved = timers["trsm_flops"] / launches / (trsm_ms * 1e-3) / 1e12
                traffic, note = pmc_traffic("f32_strip_kernel")
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
            traffic, note = pmc_traffic("step_sequential" if args.sequential else "step")
            out["roofline"] = {
                "kernel": "whole step: trsm_update_kernel<16> (dominant) + trsm_strip_kernel<*,16> + "
                          "syrk_kernel<64> + potrf_panel_fused_kernel (diagonal block + row panel), co-scheduled" if not args.sequential else
                          "whole step: trsm_strip_kernel<true,32> (dominant), then the factorisation's kernels",
                "bound": "mfma", "achieved": step_tflops, "peak": peak, "unit": "TFLOP/s",
                "frac": step_tflops / peak, "traffic": traffic, "traffic_source": note,
                "per": "step", "avg_step_ms_events": region_ms / args.steps,
                "algorithmic_flops_per_step": step_flops}
        if timers is not None and not f32:
            launches = max(1, timers["n_trsm_launches"])
            trsm_ms = timers["ms_trsm"] / launches
            achieved = timers["trsm_flops"] / launches / (trsm_ms * 1e-3) / 1e12 if trsm_ms > 0 else 0.0
            traffic, note = pmc_traffic("strip_kernel")
            out["roofline"]["isolated"] = {
                "kernel": "trsm_strip_kernel<true,32> (V = L^-1 K*, fused sum V^2 and V^T z): the sweep of every set "
                          "that is not refitted, alone on the device (instrumented pass after the timed region)",
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
        if world == 1 and args.post_steps > 0 and not f32:
            # the north-star's HBM-bound size: K(X,X) assembly for 16384 points (one fit of such a model)
            n16 = 16384
            lo_box, hi_box = np.array([b[0] for b in BOX]), np.array([b[1] for b in BOX])
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
        if world == 1 and args.post_steps > 0 and not f32:
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
