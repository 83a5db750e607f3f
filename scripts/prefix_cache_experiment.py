"""Prefix-state caching, measured (task 8).

In a last-layer parameter search (queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/mutation.py:57-59) only
the last layer's angles change between evaluations, so one could keep the state after layers 0..L-2 in a slot and apply
only the last layer per evaluation.  This script measures both ways on the device:

  full     every evaluation runs the whole L-layer circuit from |0..0> (what the library does): folded product state,
           compact first pass, fused diagonal expectation -- evaluations per second of a batched population
  cached   the last layer applied read-modify-write to a RESIDENT 2^n state (qsv_bench_ops: no folding, no compact
           tables -- the cached state is an arbitrary vector), time per application; the expectation pass a cached
           scheme also needs (one more read of the state and of the diagonal table) is NOT included, so the cached
           figure is an upper bound on what caching could reach

    python scripts/prefix_cache_experiment.py [n ...]
"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from queasars_amd import workloads  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, StatevectorDevice  # noqa: E402
from queasars_amd.evqe import EVQEPopulation  # noqa: E402
from queasars_amd.ir import CircuitIR  # noqa: E402


def bound(circuit, values):
    out = CircuitIR(circuit.n_qubits)
    for kind, target, control, theta, phi, lam in circuit.bound_ops(values):
        if kind == 1:
            out.u(theta, phi, lam, target)
        elif kind == 2:
            out.cu3(theta, phi, lam, control, target)
    return out


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [20, 24]
    for n in sizes:
        pop_size = 64 if n <= 20 else 32
        layers = 4
        population = EVQEPopulation.random_population(n, layers, pop_size, True, 0)
        # full: the last-layer search's circuits (layers 0..L-2 bound, the last one parameterised), batched
        circuits = [ind.get_partially_parameterized_quantum_circuit({layers - 1}) for ind in population.individuals]
        values = [list(ind.get_layer_parameter_values(layers - 1)) for ind in population.individuals]
        evaluator = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
        for _ in range(3):
            evaluator.evaluate_circuits(circuits, values)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            evaluator.evaluate_circuits(circuits, values)
        full_us = (time.perf_counter() - t0) / (reps * pop_size) * 1e6
        evaluator.statevector_device.close()
        # cached: only the last layer, on a resident state
        device = StatevectorDevice(n, group=1)
        times, passes = [], []
        for ind in population.individuals[:8]:
            last = CircuitIR(n)
            full = ind.get_partially_parameterized_quantum_circuit({layers - 1})
            n_prefix = len(ind.get_partially_parameterized_quantum_circuit(set()).bound_ops([])) if False else None
            # the ops of the last layer = the parameterised ones
            for row, (kind, target, control, theta, phi, lam) in zip(full._rows, full.bound_ops(list(ind.get_layer_parameter_values(layers - 1)))):
                if row[4] >= 0 or row[5] >= 0 or row[6] >= 0:
                    if kind == 1:
                        last.u(theta, phi, lam, target)
                    elif kind == 2:
                        last.cu3(theta, phi, lam, control, target)
            ms, n_passes = device.bench_ops(last, reps=20)
            times.append(ms * 1e3)
            passes.append(n_passes)
        device.close()
        print(json.dumps({
            "n": n, "layers": layers, "population": pop_size,
            "full_us_per_evaluation": round(full_us, 2), "full_evals_per_s": round(1e6 / full_us),
            "cached_last_layer_us_per_application": round(sum(times) / len(times), 2),
            "cached_passes": sum(passes) / len(passes),
            "cached_upper_bound_evals_per_s": round(1e6 / (sum(times) / len(times))),
            "note": "cached excludes the separate expectation pass it would need; full includes everything",
        }), flush=True)


if __name__ == "__main__":
    main()
