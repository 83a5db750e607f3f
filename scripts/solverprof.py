"""The EVQE driver's own time: compute_minimum_eigenvalue on the notebook's 12-qubit JSSP instance with a stand-in evaluator
that answers at once (a deterministic function of the parameter values, no device), under cProfile.  Runs without a GPU.
  python scripts/solverprof.py [population] [generations] [top]"""
import cProfile, pstats, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import jssp_instances as inst
from queasars_amd.circuit_evaluation.circuit_evaluation import BaseCircuitEvaluator
from queasars_amd.evqe.solver import (SPSA, EVQEMinimumEigensolver,
                                      EVQEMinimumEigensolverConfiguration, SPSATerminationChecker)
from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

pop = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 8
top = int(sys.argv[3]) if len(sys.argv) > 3 else 35
enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)


class Instant(BaseCircuitEvaluator):
    n_calls = 0
    n_evals = 0

    @property
    def n_qubits(self):
        return enc.n_qubits

    def evaluate_circuits(self, circuits, parameter_values):
        Instant.n_calls += 1
        Instant.n_evals += len(circuits)
        return [30.0 + float(np.cos(np.asarray(p, dtype=float)).sum()) if len(p) else 30.0 for p in parameter_values]


cfg = EVQEMinimumEigensolverConfiguration(
    optimizer=SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True,
                   termination_checker=SPSATerminationChecker(0.01, 2)),
    population_size=pop, max_generations=gens, termination_criterion=None,
    random_seed=0, n_initial_layers=2, randomize_initial_population_parameters=True,
    speciation_genetic_distance_threshold=1, use_tournament_selection=True, tournament_size=2,
    selection_alpha_penalty=0.15, selection_beta_penalty=0.02, parameter_search_probability=0.39,
    topological_search_probability=0.79, layer_removal_probability=0.02)
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(Instant())
prof.disable()
dt = time.perf_counter() - t0
print(f"{dt:.3f} s under the profiler, {Instant.n_calls} evaluator calls, {Instant.n_evals} evaluations, {result.generations} generations")
pstats.Stats(prof).sort_stats("cumulative").print_stats(top)
