"""Evaluates ONE circuit of the benchmark population (n = 20, P = 64, L = 4, seed 0) over and over: for rocprofv3 --pmc runs on a single
class of the one-launch route.    python scripts/one_circuit_loop.py <index> [repetitions]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation

index = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pop = EVQEPopulation.random_population(20, 4, 64, True, 0)
ind = pop.individuals[index]
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(20, 2020))
c, p = [ind.get_parameterized_quantum_circuit()], [list(ind.parameter_values)]
for _ in range(reps):
    ev.evaluate_circuits(c, p)
