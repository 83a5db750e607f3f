import sys, time, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
which = sys.argv[1]
if "torch" in which:
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import bench
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.distributed import evaluate_population_sharded
from queasars_amd.evqe import EVQEPopulation
from queasars_amd import workloads as helpers
n, total = 24, 256
population = EVQEPopulation.random_population(n, 4, total, True, 0)
circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
params = [list(ind.parameter_values) for ind in population.individuals]
operator = bench.ising_operator(n, 2024) if "benchop" in which else helpers.random_ising_operator(n, seed=3)
if "other" in which:
    other = OperatorCircuitEvaluator(bench.ising_operator(20, 2020))
    _, c20, p20 = helpers.population_circuits(20, 4, 64, seed=0)
    for _ in range(50): other.evaluate_circuits(c20, p20)
evaluator = OperatorCircuitEvaluator(operator)
t_end = time.perf_counter() + 0.15
while time.perf_counter() < t_end:
    evaluate_population_sharded(evaluator, circuits, params)
t0 = time.perf_counter()
for _ in range(20):
    evaluate_population_sharded(evaluator, circuits, params)
print(which, f"{(time.perf_counter() - t0) / 20 * 1e6:.0f} us per step", flush=True)
