"""EVQE driver on the CPU with an oracle-backed evaluator (the GPU run is in tests/test_gpu_parity.py).
Modelled on the reference's end-to-end test: the solver must find x = 0, y = 3 for min x^2 - y^2
(test/minimum_eigensolvers/evqe/test_evqe_algorithm.py:34-38)."""

import numpy as np
import pytest

import helpers
from queasars_amd.evqe import EVQEPopulation
from queasars_amd.evqe.solver import (
    SPSA,
    BestIndividualRelativeChangeTolerance,
    EVQEMinimumEigensolver,
    EVQEMinimumEigensolverConfiguration,
    SPSATerminationChecker,
    _SPSARun,
)
from queasars_amd.ir import PauliOperator


class OracleEvaluator:
    def __init__(self, operator):
        self.operator = operator
        self.calls = 0
        self.evaluations = 0

    @property
    def n_qubits(self):
        return self.operator.num_qubits

    def evaluate_circuits(self, circuits, parameter_values):
        self.calls += 1
        self.evaluations += len(circuits)
        return [helpers.oracle_expectation(c, p, self.operator) for c, p in zip(circuits, parameter_values)]


def xy_hamiltonian():
    return PauliOperator.from_sparse_list(
        [("Z", [0], -1.5), ("Z", [1], -3.0), ("ZZ", [0, 1], 1.0), ("Z", [2], 1.5), ("Z", [3], 3.0), ("ZZ", [2, 3], -1.0)], 4
    )


def make_config(**overrides):
    args = dict(
        optimizer=SPSA(maxiter=25, learning_rate=0.4, perturbation=0.3),
        population_size=8,
        max_generations=6,
        random_seed=0,
        n_initial_layers=2,
        randomize_initial_population_parameters=True,
        speciation_genetic_distance_threshold=2,
        use_tournament_selection=True,
        tournament_size=2,
        selection_alpha_penalty=0.1,
        selection_beta_penalty=0.1,
        parameter_search_probability=0.3,
        topological_search_probability=0.4,
        layer_removal_probability=0.05,
    )
    args.update(overrides)
    return EVQEMinimumEigensolverConfiguration(**args)


def test_solver_finds_the_ground_state_and_is_seeded():
    op = xy_hamiltonian()
    ev = OracleEvaluator(op)
    result = EVQEMinimumEigensolver(make_config()).compute_minimum_eigenvalue(ev)
    assert result.generations == 6 and len(result.best_expectation_values) == 6
    assert result.eigenvalue < -8.5  # the minimum is -9 at x = 0, y = 3
    assert sum(result.circuit_evaluations) == ev.evaluations
    # far fewer evaluator calls than evaluations: the optimiser runs are batched
    assert ev.calls * 4 < ev.evaluations
    state = helpers.oracle_state(
        result.best_individual.get_parameterized_quantum_circuit(), list(result.best_individual.parameter_values)
    )
    assert int(np.argmax(np.abs(state))) == 0b1100
    again = EVQEMinimumEigensolver(make_config()).compute_minimum_eigenvalue(OracleEvaluator(op))
    assert again.eigenvalue == result.eigenvalue and again.best_individual == result.best_individual


def test_termination_by_budget_and_by_criterion():
    op = xy_hamiltonian()
    ev = OracleEvaluator(op)
    cfg = make_config(max_generations=None, max_circuit_evaluations=1500)
    result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(ev)
    assert 0 < sum(result.circuit_evaluations) <= 1500 and result.generations >= 1
    crit = BestIndividualRelativeChangeTolerance(minimum_relative_change=0.5)
    result = EVQEMinimumEigensolver(make_config(max_generations=None, termination_criterion=crit)).compute_minimum_eigenvalue(OracleEvaluator(op))
    assert 2 <= result.generations <= 10
    with pytest.raises(ValueError):
        make_config(max_generations=None)
    with pytest.raises(ValueError):
        make_config(use_tournament_selection=True, tournament_size=None)


def test_growth_with_zero_angles_keeps_the_energy():
    """Topological search appends a layer of zero angles: identity until optimised (mutation.py:348-353)."""
    op = xy_hamiltonian()
    pop = EVQEPopulation.random_population(4, 2, 4, True, 3)
    solver = EVQEMinimumEigensolver(make_config(topological_search_probability=1.0))
    grown = solver._topological_search(pop)
    for before, after in zip(pop.individuals, grown.individuals):
        assert len(after.layers) == len(before.layers) + 1
        e0 = helpers.oracle_expectation(before.get_parameterized_quantum_circuit(), list(before.parameter_values), op)
        e1 = helpers.oracle_expectation(after.get_parameterized_quantum_circuit(), list(after.parameter_values), op)
        assert abs(e0 - e1) < 1e-13


def test_spsa_run_matches_a_plain_sequential_spsa():
    """The lock-step driver gives every run exactly the iterates of an independent SPSA with the same seed."""

    def f(x):
        return float(np.sum((x - 1.0) ** 2) + 3.0)

    cfg = SPSA(maxiter=12, learning_rate=0.2, perturbation=0.1, trust_region=True,
               termination_checker=SPSATerminationChecker(0.001, 1))
    x0 = np.array([0.3, -0.2, 2.0])
    run = _SPSARun(cfg, x0, seed=11)
    while not run.done:
        plus, minus = run.propose()
        run.accept(f(plus), f(minus))
    # plain restatement
    rng = np.random.default_rng(11)
    x, checker, nfev = x0.copy(), cfg.termination_checker.fresh(), 0
    for _ in range(cfg.maxiter):
        delta = 1 - 2 * rng.binomial(1, 0.5, size=3)
        fp, fm = f(x + 0.1 * delta), f(x - 0.1 * delta)
        nfev += 2
        update = (fp - fm) / 0.2 * delta
        norm = np.linalg.norm(update)
        if norm > 1:
            update = update / norm
        update = update * 0.2
        x = x - update
        if checker.termination_check(nfev, x, 0.5 * (fp + fm), np.linalg.norm(update), True):
            break
    assert np.array_equal(run.x, x) and run.nfev == nfev
    assert f(run.x) < f(x0)


# ---- NFT (the optimiser of the reference's own test harness, test/minimum_eigensolvers/evqe/solver.py:28-36) ----------


def test_nft_fits_a_sinusoid_exactly():
    """With every other angle fixed the expectation value is c + a cos(t - b): one NFT step lands on its minimum."""
    from queasars_amd.evqe.solver import NFT, _NFTRun

    def f(x):
        return 2.0 + 1.5 * np.cos(x[0] - 0.7) + 0.5 * np.cos(x[1] + 1.1)

    run = _NFTRun(NFT(maxfev=5), [0.2, -0.4])
    points = run.propose()
    assert len(points) == 3  # f(x), f(x + pi/2 e_0), f(x - pi/2 e_0)
    run.accept(*[f(p) for p in points])
    assert abs(np.cos(run.x[0] - 0.7) + 1.0) < 1e-12 and run.x[1] == -0.4
    points = run.propose()
    assert len(points) == 2  # the fitted minimum is reused as f(x)
    run.accept(*[f(p) for p in points])
    assert run.done and run.nfev == 5
    assert abs(f(run.x) - (2.0 - 1.5 - 0.5)) < 1e-12


def test_nft_mutation_lowers_the_sum_of_expectation_values():
    """Mirrors the reference's operator test ("sum of expectation values decreased",
    test/minimum_eigensolvers/evqe/test_evqe_operators.py:91-93) for the last-layer parameter search with NFT(maxfev=40),
    and checks the evaluation accounting; the gate angles really are sinusoidal parameters of <H>."""
    from queasars_amd.evqe.solver import NFT

    op = xy_hamiltonian()
    ev = OracleEvaluator(op)
    population = EVQEPopulation.random_population(4, 2, 6, False, 0)  # zero angles, as the reference's fixture
    solver = EVQEMinimumEigensolver(make_config(optimizer=NFT(maxfev=40), population_size=6))

    def total(pop):
        return sum(helpers.oracle_expectation(i.get_parameterized_quantum_circuit(), list(i.parameter_values), op) for i in pop.individuals)

    before = total(population)
    searched, nfev = solver._last_layer_search(ev, population)
    after = total(searched)
    assert after < before - 1.0
    assert nfev == ev.evaluations and 6 * 40 <= nfev <= 6 * 42
    # and end to end: the solver with NFT finds x = 0, y = 3 too
    result = EVQEMinimumEigensolver(make_config(optimizer=NFT(maxfev=40))).compute_minimum_eigenvalue(OracleEvaluator(op))
    assert result.eigenvalue < -8.5


def test_vectorised_spsa_driver_gives_every_run_its_own_iterates():
    """The whole-array lock-step driver (evqe/solver.py _minimize_spsa_vectorised) against the same runs advanced one by one
    through propose / accept: bitwise the same iterates and evaluation counts, for runs of different lengths, with the trust
    region both binding and not, with and without a termination checker (which stops runs at different iterations)."""
    from queasars_amd.evqe import solver as S

    class Quadratic:
        """f(x) = sum (x - 1)^2 scaled per 'circuit'; accepts whatever vectors the driver hands over (NumPy rows)."""

        def evaluate_circuits(self, circuits, parameter_values):
            return [float(c * np.sum((np.asarray(p) - 1.0) ** 2)) for c, p in zip(circuits, parameter_values)]

    rng = np.random.default_rng(3)
    for checker in (None, SPSATerminationChecker(0.05, 2)):
        for lr, pert in ((0.2, 0.1), (1.5, 0.35)):
            cfg = SPSA(maxiter=17, learning_rate=lr, perturbation=pert, trust_region=True, termination_checker=checker)
            sizes = [1, 3, 7, 12, 12, 40]
            starts = [rng.normal(size=n) for n in sizes]
            scales = [0.3, 1.0, 2.5, 0.01, 4.0, 1.0]
            ev = Quadratic()
            one_by_one = []
            for x0, scale, seed in zip(starts, scales, range(6)):
                run = _SPSARun(cfg, x0, seed=seed)
                while not run.done:
                    plus, minus = run.propose()
                    run.accept(*ev.evaluate_circuits([scale, scale], [plus, minus]))
                one_by_one.append(run)
            jobs = [(scale, _SPSARun(cfg, x0, seed=seed)) for x0, scale, seed in zip(starts, scales, range(6))]
            S._minimize_batched(ev, jobs)
            for (_, run), ref in zip(jobs, one_by_one):
                assert run.done and np.array_equal(run.x, ref.x) and run.nfev == ref.nfev and run.iteration == ref.iteration
            if checker is not None:
                assert len({run.nfev for _, run in jobs}) > 1  # (the checker stopped runs at different iterations)


# ---- the reference's operator tests (test/minimum_eigensolvers/evqe/test_evqe_operators.py), against the solver's own operators ----
# Same model (min x^2 - y^2 on four qubits), same initial population (ten individuals of two layers, seed 0), the evaluator
# backed by the oracle instead of Aer.  One deviation: the reference starts from zero angles and relies on its estimator's
# noise to leave that point (with exact values f(+eps) = f(-eps) there, SPSA's update is exactly zero); where an optimiser
# has to move, the population here starts from random angles.


def _reference_population(randomize: bool = True):
    return EVQEPopulation.random_population(4, 2, 10, randomize, 0)


def _population_values(evaluator, population):
    return evaluator.evaluate_circuits([ind.get_parameterized_quantum_circuit() for ind in population.individuals],
                                       [list(ind.parameter_values) for ind in population.individuals])


def test_reference_last_layer_search_lowers_the_expectation_values():
    """test_evqe_operators.py:64-93: the sum over the population after the mutation is below the sum before."""
    ev = OracleEvaluator(xy_hamiltonian())
    solver = EVQEMinimumEigensolver(make_config(population_size=10, optimizer=SPSA(maxiter=20, learning_rate=0.4, perturbation=0.3)))
    population = _reference_population()
    before = sum(_population_values(ev, population))
    searched, nfev = solver._last_layer_search(ev, population)
    assert nfev == 10 * 20 * 2  # (every individual, two evaluations per SPSA iteration)
    assert sum(_population_values(ev, searched)) < before
    for old, new in zip(population.individuals, searched.individuals):
        assert new.layers == old.layers  # (only parameter values move, and only the last layer's)
        keep = sum(layer.n_parameters for layer in old.layers[:-1])
        assert new.parameter_values[:keep] == old.parameter_values[:keep]


def test_reference_parameter_search_lowers_the_expectation_values():
    """test_evqe_operators.py:95-124."""
    ev = OracleEvaluator(xy_hamiltonian())
    solver = EVQEMinimumEigensolver(make_config(population_size=10, parameter_search_probability=0.3,
                                                optimizer=SPSA(maxiter=20, learning_rate=0.4, perturbation=0.3)))
    population = _reference_population()
    before = _population_values(ev, population)
    searched, nfev = solver._parameter_search(ev, population)
    after = _population_values(ev, searched)
    changed = [i for i, (a, b) in enumerate(zip(population.individuals, searched.individuals)) if a != b]
    assert changed and nfev == len(changed) * 2 * 20 * 2  # (both layers of every chosen individual)
    assert sum(after) < sum(before)
    assert all(after[i] == before[i] for i in range(10) if i not in changed)


def test_reference_topological_search_and_layer_removal_change_the_layer_count():
    """test_evqe_operators.py:126-147: with probability 0.5 topological search adds layers, layer removal takes some away."""
    population = _reference_population(randomize=False)
    layers = sum(len(ind.layers) for ind in population.individuals)
    grown = EVQEMinimumEigensolver(make_config(population_size=10, topological_search_probability=0.5))._topological_search(population)
    assert sum(len(ind.layers) for ind in grown.individuals) > layers
    shrunk = EVQEMinimumEigensolver(make_config(population_size=10, layer_removal_probability=0.5))._layer_removal(population)
    assert sum(len(ind.layers) for ind in shrunk.individuals) < layers
    assert all(len(ind.layers) >= 1 for ind in shrunk.individuals)


def test_reference_speciation_keeps_members_within_the_genetic_distance():
    """test_evqe_operators.py:149-182: after search, selection, growth and search again every member of a species is closer
    to its representative than the threshold, and the three species tables describe the whole population."""
    from queasars_amd.evqe.genome import EVQEIndividual

    distance = 2
    ev = OracleEvaluator(xy_hamiltonian())
    solver = EVQEMinimumEigensolver(make_config(population_size=10, speciation_genetic_distance_threshold=distance,
                                                topological_search_probability=1.0,
                                                optimizer=SPSA(maxiter=10, learning_rate=0.4, perturbation=0.3)))
    population, _ = solver._last_layer_search(ev, _reference_population())
    population = solver._speciation(population)
    population = solver._selection(population, _population_values(ev, population))
    population = solver._topological_search(population)
    population, _ = solver._last_layer_search(ev, population)
    population = solver._speciation(population)
    assert population.species_representatives and population.species_members and population.species_membership
    seen = set()
    for representative in population.species_representatives:
        for member in population.species_members[representative]:
            seen.add(member)
            if representative != population.individuals[member]:
                assert EVQEIndividual.get_genetic_distance(representative, population.individuals[member]) < distance
            assert population.species_membership[member] == representative
    assert seen == set(range(10))


def test_speciation_at_threshold_one_is_the_general_rule():
    """At a genetic-distance threshold of one (the notebooks') the driver finds an individual's species by hashing its layers
    (a distance below one is zero: the same layers); the result must be what the reference's rule gives compared individual by
    individual (speciation.py:34-90) -- the same representatives in the same order, the same members -- also across
    generations, where the previous generation's representatives come first."""
    from queasars_amd.evqe.genome import EVQEIndividual

    def by_the_rule(solver, population, threshold):
        representatives = list(population.species_representatives or [])
        members = {rep: [] for rep in representatives}
        for i, individual in enumerate(population.individuals):
            for rep in representatives:
                if EVQEIndividual.get_genetic_distance(individual, rep) < threshold or individual == rep:
                    members[rep].append(i)
                    break
            else:
                representatives.append(individual)
                members[individual] = [i]
        return representatives, members

    ev = OracleEvaluator(xy_hamiltonian())
    solver = EVQEMinimumEigensolver(make_config(population_size=12, speciation_genetic_distance_threshold=1, topological_search_probability=0.6,
                                                layer_removal_probability=0.2, optimizer=SPSA(maxiter=4, learning_rate=0.4, perturbation=0.3)))
    population = EVQEPopulation.random_population(4, 2, 12, True, random_seed=3)
    for _ in range(4):
        want_reps, want_members = by_the_rule(solver, population, 1)
        # (the driver then draws a new representative per species: compare the grouping, which is what the rule decides)
        groups_want = sorted(sorted(m) for m in want_members.values() if m)
        speciated = solver._speciation(population)
        groups_got = sorted(sorted(m) for m in speciated.species_members.values())
        assert groups_got == groups_want
        population = solver._selection(speciated, _population_values(ev, speciated))
        population = solver._topological_search(population)
        population = solver._layer_removal(population)


def test_reference_selection_lowers_the_expectation_values():
    """test_evqe_operators.py:184-209: three rounds of speciation and selection, the population's sum falls every round."""
    ev = OracleEvaluator(xy_hamiltonian())
    solver = EVQEMinimumEigensolver(make_config(population_size=10, optimizer=SPSA(maxiter=20, learning_rate=0.4, perturbation=0.3)))
    population, _ = solver._last_layer_search(ev, _reference_population())
    sums = [sum(_population_values(ev, population))]
    for _ in range(3):
        population = solver._speciation(population)
        population = solver._selection(population, _population_values(ev, population))
        assert len(population.individuals) == 10
        sums.append(sum(_population_values(ev, population)))
    assert all(later < earlier for earlier, later in zip(sums, sums[1:])), sums


def test_spsa_termination_checker_answers_as_the_references_does():
    """242 calls along 18 optimisations (six configurations, one checker object serving three optimisations in a row each):
    every answer, and the bookkeeping after the last call, as the reference's own class gave them
    (tests/golden/make_host_golden.py ran queasars/utility/spsa_termination.py)."""
    import json
    from pathlib import Path

    from queasars_amd.evqe.genome import new_random_seed

    data = json.loads((Path(__file__).parent / "golden" / "spsa_termination_reference.json").read_text())
    n_stop = 0
    for case in data["termination"]:
        checker = SPSATerminationChecker(case["minimum_relative_change"], case["allowed_consecutive_violations"], case["maxfev"])
        for call in case["calls"]:
            answer = checker.termination_check(call["nfev"], np.asarray(call["x"]), call["value"], call["step_size"], call["accepted"])
            assert bool(answer) == call["answer"]
            n_stop += call["answer"]
        after = case["after"]
        assert checker.n_function_evaluations == after["n_function_evaluations"]
        assert checker.function_value_history == after["function_value_history"]
        assert checker.n_function_evaluation_history == after["n_function_evaluation_history"]
        assert checker.best_function_value == after["best_function_value"]
        if after["best_parameter_values"] is None:
            with pytest.raises(ValueError):
                checker.best_parameter_values
        else:
            assert list(checker.best_parameter_values) == after["best_parameter_values"]
    assert n_stop >= 20
    from random import Random

    for seed, chain in data["seed_chains"].items():
        rng = Random(int(seed))
        assert [new_random_seed(rng) for _ in chain] == chain


def test_search_on_shared_fully_parameterised_circuits_is_the_search_on_bound_ones(monkeypatch):
    """The driver searches a layer inside the individual's fully parameterised circuit (one shared object per structure, the
    other layers' values as parameter values that do not move) where the reference binds them into a fresh circuit: the same
    matrices gate for gate, so the same run -- eigenvalue, best individual, evaluations per generation --, with far fewer
    circuit objects handed to the evaluator."""
    op = xy_hamiltonian()

    class Counting(OracleEvaluator):
        def __init__(self, operator):
            super().__init__(operator)
            self.seen = set()

        def evaluate_circuits(self, circuits, parameter_values):
            self.seen.update(id(c) for c in circuits)
            self.keep = getattr(self, "keep", []) + list(circuits)  # (ids stay unique while the objects live)
            return super().evaluate_circuits(circuits, parameter_values)

    results = {}
    for share in ("2", "0"):  # ("2": embedded also where the host packs the points, as here)
        monkeypatch.setenv("QSV_SHARE_CIRCUITS", share)
        ev = Counting(op)
        results[share] = (EVQEMinimumEigensolver(make_config()).compute_minimum_eigenvalue(ev), len(ev.seen))
    shared, bound = results["2"][0], results["0"][0]
    assert shared.eigenvalue == bound.eigenvalue and shared.best_individual == bound.best_individual
    assert shared.circuit_evaluations == bound.circuit_evaluations
    assert shared.best_expectation_values == bound.best_expectation_values
    assert results["2"][1] * 3 < results["0"][1]  # (distinct circuit objects the evaluator was shown)
    # the same for the run-by-run driver
    monkeypatch.setenv("QSV_SCALAR_SPSA", "1")
    monkeypatch.setenv("QSV_SHARE_CIRCUITS", "2")
    scalar = EVQEMinimumEigensolver(make_config()).compute_minimum_eigenvalue(OracleEvaluator(op))
    assert scalar.eigenvalue == shared.eigenvalue and scalar.circuit_evaluations == shared.circuit_evaluations


def test_embedded_search_of_an_individual_with_more_than_ten_layers(monkeypatch):
    """The fully parameterised circuit numbers its parameters in name-sorted block order (layer10_ before layer2_), the
    individual keeps its values in layer order: from eleven layers on the two differ, and the embedded search has to place
    the searched layer's point -- and every other layer's values -- by the circuit's order.  Every point of the embedded
    form gives the ops of the bound circuit (mutation.py:57-59: a layer alone, the others bound), and the two searches
    return the same individuals."""
    from queasars_amd.evqe.genome import EVQEIndividual
    from queasars_amd.evqe.solver import _full_point

    individual = EVQEIndividual.random_individual(4, 12, True, random_seed=5)
    full = individual.get_parameterized_quantum_circuit(shared=True)
    base = np.asarray(individual.parameter_values_in_circuit_order())
    assert sorted(base.tolist()) == sorted(individual.parameter_values) and base.tolist() != list(individual.parameter_values)
    for layer in (5, 10, 1, 11, -1):
        layer %= 12
        bound = individual.get_partially_parameterized_quantum_circuit({layer})
        start = individual.circuit_parameter_offsets[layer]
        positions = np.arange(start, start + individual.layers[layer].n_parameters)

        class Run:
            embed = (base, positions)

        point = np.linspace(0.1, 0.9, positions.size)
        assert full.bound_ops(_full_point(Run, point).tolist()) == bound.bound_ops(point.tolist())
    # the individual's own values at its own layer: the bound circuit again
    assert full.bound_ops(base.tolist()) == individual.get_partially_parameterized_quantum_circuit(set()).bound_ops([])

    op = xy_hamiltonian()
    solver_results = {}
    for share in ("2", "0"):
        monkeypatch.setenv("QSV_SHARE_CIRCUITS", share)
        solver = EVQEMinimumEigensolver(make_config(optimizer=SPSA(maxiter=6, learning_rate=0.4, perturbation=0.3)))
        individuals = [EVQEIndividual.random_individual(4, 12, True, random_seed=s) for s in (5, 6, 7)]
        solver_results[share] = solver._optimize_layers(OracleEvaluator(op), individuals, [5, 10, -1], [11, 12, 13])
    assert solver_results["2"][1] == solver_results["0"][1]
    for a, b in zip(solver_results["2"][0], solver_results["0"][0]):
        assert a == b


class _RoundedEvaluator(OracleEvaluator):
    """An evaluator whose values carry single-precision error: what a search evaluator is allowed to be."""

    def evaluate_circuits(self, circuits, parameter_values):
        return [float(np.float32(v)) for v in super().evaluate_circuits(circuits, parameter_values)]


def test_a_search_evaluator_serves_the_searches_and_only_them():
    """``compute_minimum_eigenvalue(evaluator, search_evaluator)``: the optimiser runs evaluate on the second evaluator (a
    single-precision handle), every fitness value -- what selection compares and the result reports -- comes from the first."""
    op = xy_hamiltonian()
    fitness, search = OracleEvaluator(op), _RoundedEvaluator(op)
    result = EVQEMinimumEigensolver(make_config()).compute_minimum_eigenvalue(fitness, search)
    assert result.eigenvalue < -8.5 and result.generations == 6
    # one fitness call per generation, the whole population each (possibly shrunk by speciation); everything else on the searcher
    assert fitness.calls == 6 and fitness.evaluations <= 6 * 8
    assert search.evaluations + fitness.evaluations == sum(result.circuit_evaluations)
    assert search.evaluations > 10 * fitness.evaluations
    exact = helpers.oracle_expectation(result.best_individual.get_parameterized_quantum_circuit(),
                                       list(result.best_individual.parameter_values), op)
    assert result.eigenvalue == exact  # (not a rounded value)
    with pytest.raises(ValueError):
        EVQEMinimumEigensolver(make_config()).compute_minimum_eigenvalue(
            fitness, OracleEvaluator(PauliOperator.from_sparse_list([("Z", [0], 1.0)], 5)))
