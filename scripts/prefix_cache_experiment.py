"""Kept states in a layer search, measured.

In a layer search (queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/mutation.py:57-59) only one layer's angles
change between evaluations.  Two ways to evaluate the search's points, both batched over the population, expectation value
included, whole evaluator calls:

  full    every evaluation runs the whole circuit (the other layers bound) from |0..0>: what rounds 1-3 did
  kept    the state in front of the searched layer is computed ONCE per individual and kept on the device
          (qsv_prefix_create); every evaluation applies the searched layer and what follows it to that state
          (qsv_circuits_create_on_prefixes)

Round 2's version of this script timed one unbatched cached pass without its expectation pass, at four layers only.

    python scripts/prefix_cache_experiment.py [n:layers:population ...]     (default 20:8:64 20:7:64 20:6:64 24:8:32)
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from queasars_amd import workloads  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator  # noqa: E402
from queasars_amd.evqe import EVQEPopulation  # noqa: E402


def rate(evaluator, circuits, values, seconds=0.4):
    for _ in range(3):
        out = evaluator.evaluate_circuits(circuits, values)
    t0 = time.perf_counter()
    evaluator.evaluate_circuits(circuits, values)
    per = time.perf_counter() - t0
    reps = max(3, int(seconds / max(per, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(reps):
        evaluator.evaluate_circuits(circuits, values)
    return len(circuits) * reps / (time.perf_counter() - t0), np.asarray(out)


def main():
    specs = sys.argv[1:] or ["20:8:64", "20:7:64", "20:6:64", "24:8:32"]
    for spec in specs:
        n, layers, pop = (int(x) for x in spec.split(":"))
        population = EVQEPopulation.random_population(n, layers, pop, True, 0)
        evaluator = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020 if n == 20 else 2024))
        only_kept = os.environ.get("QSV_EXP_ONLY") == "kept"  # (profiling runs: the last-layer search's kept-state evaluations alone)
        for layer in ([layers - 1] if only_kept else sorted({layers - 1, layers // 2}, reverse=True)):
            if only_kept:
                pairs = [ind.get_layer_search_circuits(layer) for ind in population.individuals]
                values = [list(ind.get_layer_parameter_values(layer)) for ind in population.individuals]
                states = evaluator.keep_states([p for p, _ in pairs], [[] for _ in pairs])
                kept = [s.continue_from(state) for (_, s), state in zip(pairs, states)]
                kept_rate, _ = rate(evaluator, kept, values, seconds=0.1)
                costs = evaluator.circuit_costs(kept)
                print(json.dumps({"n": n, "layers": layers, "population": pop, "kept_evals_per_s": round(kept_rate),
                                  "passes": [c["n_passes"] for c in costs]}), flush=True)
                continue
            full = [ind.get_partially_parameterized_quantum_circuit({layer}) for ind in population.individuals]
            values = [list(ind.get_layer_parameter_values(layer)) for ind in population.individuals]
            costs_full = evaluator.circuit_costs(full)
            full_rate, full_values = rate(evaluator, full, values)
            pairs = [ind.get_layer_search_circuits(layer) for ind in population.individuals]
            t0 = time.perf_counter()
            states = evaluator.keep_states([p for p, _ in pairs], [[] for _ in pairs])
            kept = [s.continue_from(state) for (_, s), state in zip(pairs, states)]
            evaluator.evaluate_circuits(kept, values)  # (registration of the continued circuits)
            setup_ms = (time.perf_counter() - t0) * 1e3
            costs_kept = evaluator.circuit_costs(kept)
            kept_rate, kept_values = rate(evaluator, kept, values)
            # the population as a search would mix it: kept states only where the circuit has no split form
            mixed = [k if c["route"] == "gate passes" else f for k, f, c in zip(kept, full, costs_full)]
            mixed_rate, mixed_values = rate(evaluator, mixed, values)
            print(json.dumps({
                "n": n, "layers": layers, "population": pop, "searched_layer": layer,
                "full_evals_per_s": round(full_rate), "kept_evals_per_s": round(kept_rate), "mixed_evals_per_s": round(mixed_rate),
                "gain_kept": round(kept_rate / full_rate, 2), "gain_mixed": round(mixed_rate / full_rate, 2),
                "max_abs_diff_kept_vs_full": float(np.abs(kept_values - full_values).max()),
                "max_abs_diff_mixed_vs_full": float(np.abs(mixed_values - full_values).max()),
                "routes_full": {r: sum(c["route"] == r for c in costs_full) for r in {c["route"] for c in costs_full}},
                "passes_full_mean": float(np.mean([c["n_passes"] for c in costs_full if c["route"] == "gate passes"] or [0])),
                "passes_kept_mean": float(np.mean([c["n_passes"] for c in costs_kept])),
                "setup_ms_states_and_registration": round(setup_ms, 2),
                "kept_states_alive": evaluator.statevector_device.kept_state_count(),
            }), flush=True)
            del states, kept, mixed, pairs
        evaluator.statevector_device.close()


if __name__ == "__main__":
    main()
