"""Per-step wall time of the benchmark population by where its parameter values live: Python lists, NumPy rows, a matrix in
device memory (with and without the ready event).  python scripts/devparams_time.py [n] [layers] [population]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, L, P = (int(x) for x in (sys.argv[1:4] + ["20", "4", "64"][len(sys.argv[1:4]):]))
_, circuits, params = workloads.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
width = max(len(p) for p in params)
host = np.zeros((P, width))
for i, p in enumerate(params):
    host[i, : len(p)] = p
rows = [host[i, : len(p)].copy() for i, p in enumerate(params)]
matrix = torch.from_numpy(host).cuda()
torch.cuda.synchronize()
want = ev.evaluate_circuits(circuits, params)
assert ev.evaluate_circuits(circuits, matrix) == want


def timed(fn, reps=4000):
    for _ in range(300):
        fn()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


print(f"n = {n}, {L} layers, {P} individuals, {sum(len(p) for p in params)} parameter values (us per step, best of three runs of 4000)")
print("  host lists of floats          %7.2f" % timed(lambda: ev.evaluate_circuits(circuits, params)))
print("  NumPy rows                    %7.2f" % timed(lambda: ev.evaluate_circuits(circuits, rows)))
print("  device matrix, ready event    %7.2f" % timed(lambda: ev.evaluate_circuits(circuits, matrix)))
print("  device matrix, known complete %7.2f" % timed(lambda: ev.evaluate_device_parameters(circuits, matrix, ready=True)))
