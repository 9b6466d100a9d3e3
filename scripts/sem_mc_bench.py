"""Monte-Carlo target (f4): device throughput on the complete graph, do(D, E) over a grid, 100 000 draws each,
against the stacked numpy restatement (oracle/sem_oracle.py) and the reference-style Python loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cbo_with_oop_amd.graphs import CompleteGraph
from oracle import sem_oracle as S

dev = CompleteGraph.define_sem().device()
for side in (1, 32, 128):
    d, e = np.meshgrid(np.linspace(-5, 5, side), np.linspace(-6, 3, side), indexing="ij")
    vals = np.stack([d.ravel(), e.ravel()], axis=1)
    dev.target_means(["D", "E"], vals)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): dev.target_means(["D", "E"], vals)
    dt = (time.perf_counter() - t0) / reps
    print(f"device: {len(vals):6d} interventions x 100000 draws: {dt*1e3:9.3f} ms  -> {len(vals)*1e5/dt/1e9:8.2f} G draws/s", flush=True)
sem = S.complete_graph_sem()
t0 = time.perf_counter(); S.compute_interventions(sem, {"D": 1.0, "E": 0.5}); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); S.compute_interventions_loop(sem, {"D": 1.0, "E": 0.5}, num_samples=5000); t2 = (time.perf_counter() - t0) * 20
print(f"numpy stacked restatement: {t1*1e3:.1f} ms per intervention; reference-style Python loop: {t2:.2f} s per intervention (5000 draws x20)")
