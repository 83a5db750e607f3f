"""BASELINE config 4 end to end: EVQE (SPSA, 33 iterations) on the notebook's 12-qubit JSSP instance, sampler branch (512
shots, CVaR 0.5) and estimator branch, population 10 (the notebook's) and 64; one JSON line per run.
QSV_SCALAR_SPSA=1: the run-by-run optimiser loop instead of the whole-array one (evqe/solver.py)."""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import jssp_instances as inst
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, OperatorSamplerCircuitEvaluator
from queasars_amd.evqe.solver import (SPSA, BestIndividualRelativeChangeTolerance, EVQEMinimumEigensolver,
                                      EVQEMinimumEigensolverConfiguration, SPSATerminationChecker)
from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
op = enc.get_problem_hamiltonian()
for branch in ("sampler", "estimator"):
    for pop in (10, 64):
        for seed in (0, 1):
            evaluator = OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=seed) if branch == "sampler" else OperatorCircuitEvaluator(op)
            cfg = EVQEMinimumEigensolverConfiguration(
                optimizer=SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True,
                               termination_checker=SPSATerminationChecker(0.01, 2)),
                population_size=pop, max_generations=8, termination_criterion=BestIndividualRelativeChangeTolerance(0.01, 1),
                random_seed=seed, n_initial_layers=2, randomize_initial_population_parameters=True,
                speciation_genetic_distance_threshold=1, use_tournament_selection=True, tournament_size=2,
                selection_alpha_penalty=0.15, selection_beta_penalty=0.02, parameter_search_probability=0.39,
                topological_search_probability=0.79, layer_removal_probability=0.02)
            t0 = time.perf_counter()
            result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(evaluator)
            dt = time.perf_counter() - t0
            best = result.best_individual
            probs = evaluator.statevector_device.probabilities(best.get_parameterized_quantum_circuit(), list(best.parameter_values))
            schedule = enc.translate_result_bitstring(format(int(np.argmax(probs)), f"0{enc.n_qubits}b"))
            print(json.dumps({"instance": "notebook 2x3", "branch": branch, "population": pop, "seed": seed,
                              "optimiser_loop": "run by run" if os.environ.get("QSV_SCALAR_SPSA") else "whole-array",
                              "eigenvalue": result.eigenvalue, "generations": result.generations,
                              "circuit_evaluations": int(sum(result.circuit_evaluations)),
                              "evals_per_s": round(sum(result.circuit_evaluations) / dt),
                              "seconds": round(dt, 4), "top_state_valid": bool(schedule.is_valid), "top_state_makespan": schedule.makespan}), flush=True)
            evaluator.statevector_device.close()
