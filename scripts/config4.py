"""BASELINE config 4: the JSSP instance of examples/evqe_jssp_optimization.ipynb solved end to end by EVQE on one GPU
(sampler + CVaR(0.5), 512 shots, SPSA 33 iterations, population 10, 2 initial layers, tournament size 2), plus a
3 x 3 instance (18 qubits)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

import jssp_instances as inst  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, OperatorSamplerCircuitEvaluator  # noqa: E402
from queasars_amd.evqe.solver import (  # noqa: E402
    SPSA, BestIndividualRelativeChangeTolerance, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration, SPSATerminationChecker,
)
from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder  # noqa: E402


def solve(encoder, branch, seed, max_generations):
    op = encoder.get_problem_hamiltonian()
    if branch == "sampler":
        evaluator = OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=seed)
    else:
        evaluator = OperatorCircuitEvaluator(op)
    cfg = EVQEMinimumEigensolverConfiguration(
        optimizer=SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True,
                       termination_checker=SPSATerminationChecker(0.01, 2)),
        population_size=10, max_generations=max_generations,
        termination_criterion=BestIndividualRelativeChangeTolerance(0.01, 1), random_seed=seed, n_initial_layers=2,
        randomize_initial_population_parameters=True, speciation_genetic_distance_threshold=1,
        use_tournament_selection=True, tournament_size=2, selection_alpha_penalty=0.15, selection_beta_penalty=0.02,
        parameter_search_probability=0.39, topological_search_probability=0.79, layer_removal_probability=0.02,
    )
    t0 = time.perf_counter()
    result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(evaluator)
    dt = time.perf_counter() - t0
    best = result.best_individual
    probs = evaluator._device.probabilities(best.get_parameterized_quantum_circuit(), list(best.parameter_values))
    top = int(np.argmax(probs))
    schedule = encoder.translate_result_bitstring(format(top, f"0{encoder.n_qubits}b"))
    evals = sum(result.circuit_evaluations)
    return {
        "branch": branch, "seed": seed, "n_qubits": encoder.n_qubits, "generations": result.generations,
        "best_per_generation": [round(v, 4) for v in result.best_expectation_values], "eigenvalue": result.eigenvalue,
        "circuit_evaluations": evals, "seconds": round(dt, 3), "evals_per_s": round(evals / dt, 1),
        "top_state_probability": float(probs[top]), "top_state_valid": schedule.is_valid, "top_state_makespan": schedule.makespan,
        "layers_of_best": len(best.layers),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0,1,2")
    ap.add_argument("--max-generations", type=int, default=8)
    args = ap.parse_args()
    enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
    for seed in (int(s) for s in args.seeds.split(",")):
        for branch in ("sampler", "estimator"):
            print(json.dumps({"instance": "notebook 2x3", **solve(enc, branch, seed, args.max_generations)}), flush=True)
    enc3 = JSSPDomainWallHamiltonianEncoder(inst.three_by_three(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
    print(json.dumps({"instance": "3x3 (18 qubits)", **solve(enc3, "sampler", 0, args.max_generations)}), flush=True)


if __name__ == "__main__":
    main()
