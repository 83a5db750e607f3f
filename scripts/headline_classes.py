"""Which circuits set the length of the headline's one launch?  The benchmark population (n = 20, P = 64, L = 4) evaluated class by
class -- by number of cut keys of each circuit's split form -- with HIP events around the launch (device.profile()).
  python scripts/headline_classes.py"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation

n, P, L = 20, 64, 4
pop = EVQEPopulation.random_population(n, L, P, True, 0)
circuits = [ind.get_parameterized_quantum_circuit() for ind in pop.individuals]
params = [list(ind.parameter_values) for ind in pop.individuals]
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, 2020))
dev = ev.statevector_device
costs = ev.circuit_costs(circuits)
classes = {}
for i, c in enumerate(costs):
    classes.setdefault((c["route"], c["n_keys"]), []).append(i)

def launch_us(idx, reps=30):
    cs, ps = [circuits[i] for i in idx], [params[i] for i in idx]
    for _ in range(5):
        ev.evaluate_circuits(cs, ps)
    t0 = time.perf_counter()
    for _ in range(reps):
        ev.evaluate_circuits(cs, ps)
    wall = (time.perf_counter() - t0) / reps * 1e6
    dev.set_profiling(True)
    acc, launches = 0.0, 0
    for _ in range(reps):
        ev.evaluate_circuits(cs, ps)
        p = dev.profile()
        acc += p["kernel_ms"][0]; launches += p["kernel_launches"][0]
    dev.set_profiling(False)
    return wall, acc / max(1, launches) * 1e3, launches / reps

print("all 64:", launch_us(list(range(P))))
for key, idx in sorted(classes.items()):
    print(key, len(idx), "circuits:", launch_us(idx), " one alone:", launch_us(idx[:1]), " eight:", launch_us((idx * 8)[:8]))
