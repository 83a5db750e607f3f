"""One rank's step of the sharded evaluation on ONE GPU (nccl group of one rank): evaluation alone, evaluation + the
fitness all-gather path as evaluate_population_sharded runs it for world > 1."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import torch.distributed as dist
from queasars_amd import distributed as qd
from queasars_amd import workloads as helpers
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
n, P = 20, 64
_, circuits, params = helpers.population_circuits(n, 4, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=3))
dev = torch.device("cuda", 0)

def timed(fn, reps=300):
    for _ in range(30):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps * 1e6, out

t_eval, want = timed(lambda: ev.evaluate_circuits(circuits, params))
t_both, got = timed(lambda: qd._gather(ev.evaluate_circuits(circuits, params), P, 1, 0, None, dev))
assert list(got) == list(want)
print(f"evaluation {t_eval:.1f} us; evaluation + gather path {t_both:.1f} us")
os.environ["QSV_GATHER_NODE"] = "0"  # (first the collective's ways; the node's shared table at the end)
if hasattr(qd, "evaluate_block_and_gather"):
    # round 4: the collective receives into host memory the device addresses, the end is read off the slots (QSV_GATHER_HOST=0:
    # round 3's copy back + stream synchronisation)
    for host in ("0", "1", "0", "1"):
        os.environ["QSV_GATHER_HOST"] = host
        t_fused, got2 = timed(lambda: qd.evaluate_block_and_gather(ev, circuits, params, P, 1, 0, None, dev))
        assert list(got2) == list(want), (got2[:3], want[:3])
        state = qd._chain_state(ev, dev)
        print(f"chained step, receive {'into host-mapped memory, polled' if host == '1' else 'on the device, copied back'}: "
              f"{t_fused:.1f} us (+{t_fused - t_eval:.1f} over the evaluation alone; host receive refused: {bool(state.get('no_host_receive'))})")
    # the same with the parameter values resident in device memory (what bench.py feeds)
    import numpy as np
    width = max(len(p) for p in params)
    m = np.zeros((P, width))
    for i, p in enumerate(params):
        m[i, : len(p)] = p
    matrix = torch.from_numpy(m).cuda()
    torch.cuda.synchronize()
    t_eval_dev, _ = timed(lambda: ev.evaluate_circuits(circuits, matrix))
    for host in ("0", "1"):
        os.environ["QSV_GATHER_HOST"] = host
        t_fused, got3 = timed(lambda: qd.evaluate_block_and_gather(ev, circuits, matrix, P, 1, 0, None, dev))
        assert list(got3) == list(want)
        print(f"device-resident inputs: evaluation {t_eval_dev:.1f} us; chained step (host receive {host}) {t_fused:.1f} us (+{t_fused - t_eval_dev:.1f})")
    # the node's shared table (no collective: the kernels store into the rank's slot, the host reads the table)
    os.environ["QSV_GATHER_NODE"] = "1"
    for values, name, base in ((params, "host lists", t_eval), (matrix, "device-resident inputs", t_eval_dev)):
        t_node, got4 = timed(lambda: qd.evaluate_block_and_gather(ev, circuits, values, P, 1, 0, None, dev))
        assert list(got4) == list(want)
        table = qd._node_table(None, 1, 0, dev)
        print(f"{name}: step through the node's shared table {t_node:.1f} us (+{t_node - base:.1f} over the evaluation alone; "
              f"table registered with the GPU: {bool(table is not None and table.registered)}, steps through it: {table.step if table else 0})")
dist.destroy_process_group()
