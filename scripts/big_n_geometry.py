"""Whole evaluations through the 2^n state at n >= 26 (states beyond the Infinity Cache) under the environment's geometry:
  python scripts/big_n_geometry.py n layers population precision"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation
n, layers, pop, precision = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
population = EVQEPopulation.random_population(n, layers, pop, True, 0)
circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
params = [list(ind.parameter_values) for ind in population.individuals]
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, 2024), dtype=precision)
ev.statevector_device.set_option("split", 0)
for _ in range(2):
    ev.evaluate_circuits(circuits, params)
dev = ev.statevector_device
dev.set_profiling(True)
ev.evaluate_circuits(circuits, params)
prof = dev.profile()
dev.set_profiling(False)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    ev.evaluate_circuits(circuits, params)
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"n": n, "layers": layers, "precision": precision, "evals_per_s": round(pop / dt, 2), "state_passes": prof["n_state_passes"],
                  "later_pass_ms_total": round(prof["kernel_ms"][1], 2), "later_pass_bytes": prof["kernel_bytes"][1],
                  "later_frac_hbm": round(prof["kernel_bytes"][1] / (prof["kernel_ms"][1] * 1e-3) / 8e12, 3)}))
