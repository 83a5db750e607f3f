import sys, time
sys.path.insert(0, "/root/repo")
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
n, L, P = 20, 4, 64
_, circuits, params = workloads.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
for _ in range(200): ev.evaluate_circuits(circuits, params)
t0 = time.perf_counter()
N = 4000
for _ in range(N): ev.evaluate_circuits(circuits, params)
print("whole call %.2f us" % ((time.perf_counter() - t0) / N * 1e6))
