"""Cost of the fitness all-gather path of evaluate_population_sharded on ONE GPU (nccl process group of one rank): the
staging copies, the collective's launch and the final synchronisation -- everything except the other ranks."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import torch.distributed as dist
from queasars_amd import distributed as qd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
local = list(np.random.default_rng(0).normal(size=64))
dev = torch.device("cuda", 0)
for _ in range(20):
    qd._gather(local, 64, 1, 0, None, dev)
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    out = qd._gather(local, 64, 1, 0, None, dev)
dt = (time.perf_counter() - t0) / reps
assert out == local
print(f"gather path (1 rank, nccl): {dt * 1e6:.1f} us per call")
dist.destroy_process_group()
