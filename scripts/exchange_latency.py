"""Latency of the per-step arg-max exchange (torch.distributed all-gather of one 16-byte record per rank) next to the
library's own streams.  Launch with torch.distributed.run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from cbo_with_oop_amd import _lib
from cbo_with_oop_amd.sharding import exchange_argmax
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
ctx = _lib.Context.get(torch.cuda.current_device())
for dev in (None, torch.device("cpu")) if backend != "nccl" else (None,):
    for _ in range(20): exchange_argmax(1.0 + rank, 7 + rank, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): exchange_argmax(1.0 + rank, 7 + rank, device=dev)
    dt = (time.perf_counter() - t0) / 200
    if rank == 0: print(f"backend {backend} device {dev}: exchange_argmax {dt*1e6:.0f} us per call, world {world}", flush=True)
dist.barrier(); dist.destroy_process_group()
