"""Pieces of the chained evaluate + all-gather step on ONE GPU (nccl group of one rank)."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import torch.distributed as dist
from queasars_amd import distributed as qd
from queasars_amd import workloads as helpers
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
n, P = 20, 64
_, circuits, params = helpers.population_circuits(n, 4, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=3))
dev = torch.device("cuda", 0)
_, send, recv, recv_host = qd._buffers(1, P, dev)

def timed(fn, reps=300):
    for _ in range(30):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6

print(f"evaluate_circuits (library's own stream): {timed(lambda: ev.evaluate_circuits(circuits, params)):.1f} us")
stream = qd._chain_stream(ev, dev)
print(f"evaluate_circuits (torch stream):          {timed(lambda: ev.evaluate_circuits(circuits, params)):.1f} us")

def v2():
    ev.evaluate_circuits_to_device(circuits, params, send.data_ptr())
    stream.synchronize()
print(f"to_device + synchronize:                   {timed(v2):.1f} us")

def v2c():
    ev.evaluate_circuits_to_device(circuits, params, send.data_ptr())
    torch.cuda.synchronize()
print(f"to_device + torch.cuda.synchronize():      {timed(v2c):.1f} us")

def v2d():
    ev.evaluate_circuits_to_device(circuits, params, send.data_ptr())
    ev.statevector_device._lib.qsv_set_profiling(ev.statevector_device._handle, 0)  # (takes the handle: no wait by itself)
    stream.synchronize()
print(f"to_device + a handle call + synchronize:   {timed(v2d):.1f} us")

def v2b():
    with torch.cuda.stream(stream):
        ev.evaluate_circuits_to_device(circuits, params, send.data_ptr())
    stream.synchronize()
print(f"  ... inside torch.cuda.stream():          {timed(v2b):.1f} us")

def v3():
    with torch.cuda.stream(stream):
        ev.evaluate_circuits_to_device(circuits, params, send.data_ptr())
        dist.all_gather_into_tensor(recv, send)
    stream.synchronize()
print(f"to_device + all_gather + synchronize:      {timed(v3):.1f} us")

def v4():
    with torch.cuda.stream(stream):
        ev.evaluate_circuits_to_device(circuits, params, send.data_ptr())
        recv_host.copy_(send, non_blocking=True)
    stream.synchronize()
print(f"to_device + copy to host + synchronize:    {timed(v4):.1f} us")

def v5():
    with torch.cuda.stream(stream):
        dist.all_gather_into_tensor(recv, send)
    stream.synchronize()
print(f"all_gather + synchronize alone:            {timed(v5):.1f} us")
dist.destroy_process_group()
