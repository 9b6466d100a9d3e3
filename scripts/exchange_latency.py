"""Latency of the per-step arg-max exchange (cbo_comm_argmax: one ncclAllGather of a 16-byte record per rank on the
communicator's own stream, issued by the library) -- single process here, or under any one-process-per-GPU launcher
that sets RANK / WORLD_SIZE / LOCAL_RANK."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbo_with_oop_amd import _lib
from cbo_with_oop_amd.sharding import Communicator

comm = Communicator.from_env() or Communicator.single(_lib.Context.get())
rank, world = comm.rank, comm.world
for _ in range(20):
    comm.argmax(1.0 + rank, 7 + rank)
comm.barrier()
t0 = time.perf_counter()
for _ in range(200):
    comm.argmax(1.0 + rank, 7 + rank)
dt = (time.perf_counter() - t0) / 200
slowest = comm.max(dt)
if rank == 0:
    print(f"cbo_comm_argmax: {slowest*1e6:.0f} us per call (slowest rank), world {world}", flush=True)
comm.close()
