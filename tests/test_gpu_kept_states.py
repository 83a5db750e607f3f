"""GPU tests of kept states (include/qsv.h: qsv_prefix_create, qsv_circuits_create_on_prefixes) and of the layer searches that
use them.  Reference behaviour: a layer search evaluates get_partially_parameterized_quantum_circuit({layer_id}) -- every other
layer bound -- again and again (queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/mutation.py:57-59), so the state in
front of the searched layer is the same in every evaluation.  fp64 values within 1e-10 of the oracle (north_star's tolerance),
amplitudes within 1e-12; fp32 within 2e-6 * sum |c_k| of fp64.  Run on the MI355X box with -m gpu."""

import ctypes as C
import gc
import pickle

import numpy as np
import pytest

import helpers
from queasars_amd import _lib
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, OperatorSamplerCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation

pytestmark = pytest.mark.gpu

EXP_TOL = 1e-10


def search_circuits(population, layer):
    """Per individual: (whole circuit with `layer` free, the layers in front bound, the rest on top, the layer's values)."""
    out = []
    for ind in population.individuals:
        front, rest = ind.get_layer_search_circuits(layer)
        out.append((ind.get_partially_parameterized_quantum_circuit({layer}), front, rest, list(ind.get_layer_parameter_values(layer))))
    return out


@pytest.mark.parametrize("n,layers,layer", [(14, 6, 5), (14, 6, 2), (16, 5, 4), (13, 4, 1), (10, 4, 3), (12, 5, 2)])
def test_circuits_on_kept_states_against_the_oracle(n, layers, layer):
    """Values and amplitudes of circuits that continue a kept state: the NumPy oracle's for the whole circuit."""
    population = EVQEPopulation.random_population(n, layers, 8, True, 3)
    op = helpers.random_ising_operator(n, seed=n)
    evaluator = OperatorCircuitEvaluator(op)
    cases = search_circuits(population, layer)
    states = evaluator.keep_states([front for _, front, _, _ in cases], [[] for _ in cases])
    kept = [rest.continue_from(state) for (_, _, rest, _), state in zip(cases, states)]
    values = [v for _, _, _, v in cases]
    got = np.asarray(evaluator.evaluate_circuits(kept, values))
    ref = np.asarray([helpers.oracle_expectation(whole, v, op) for (whole, _, _, v) in cases])
    assert np.abs(got - ref).max() < EXP_TOL
    # other points of the search: the kept state does not move
    rng = np.random.default_rng(5)
    for _ in range(3):
        points = [list(rng.uniform(0, 2 * np.pi, len(v))) for v in values]
        got = np.asarray(evaluator.evaluate_circuits(kept, points))
        ref = np.asarray([helpers.oracle_expectation(whole, p, op) for (whole, _, _, _), p in zip(cases, points)])
        assert np.abs(got - ref).max() < EXP_TOL
    # the state itself
    dev = evaluator.statevector_device
    for (whole, _, _, v), circuit in list(zip(cases, kept))[:3]:
        assert np.abs(dev.statevector(circuit, v) - helpers.oracle_state(whole, v)).max() < 1e-12
    # general operator (the state is stored and the grouped expectation kernel reads it)
    general = helpers.random_pauli_operator(n, 12, seed=7)
    ev2 = OperatorCircuitEvaluator(general, statevector_device=dev)
    got = np.asarray(ev2.evaluate_circuits(kept, values))
    ref = np.asarray([helpers.oracle_expectation(whole, v, general) for (whole, _, _, v) in cases])
    assert np.abs(got - ref).max() < EXP_TOL


def test_kept_and_whole_circuits_in_one_batch_and_the_same_bits_whatever_the_batch():
    """A search's batch mixes individuals with a split form (whole circuits), unsplittable ones evaluated whole, and ones on
    kept states -- pushed in one piece or several, on one stream or two, from lists or from a matrix in device memory: every
    evaluation's value is the same bits in any company."""
    import torch

    n, layers = 15, 6
    population = EVQEPopulation.random_population(n, layers, 24, True, 11)
    op = helpers.random_ising_operator(n, seed=15)
    evaluator = OperatorCircuitEvaluator(op)
    dev = evaluator.statevector_device
    cases = search_circuits(population, layers - 1)
    whole = [c[0] for c in cases]
    values = [c[3] for c in cases]
    costs = evaluator.circuit_costs(whole)
    routes = {c["route"] for c in costs}
    assert "gate passes" in routes and len(routes) >= 2, routes  # (the population is of mixed depth on this device)
    states = evaluator.keep_states([c[1] for c in cases], [[] for _ in cases])
    kept = [c[2].continue_from(s) for c, s in zip(cases, states)]
    # every second unsplittable individual on its kept state, the others whole
    mixed, flip = [], True
    for w, k, cost in zip(whole, kept, costs):
        if cost["route"] == "gate passes":
            mixed.append(k if flip else w)
            flip = not flip
        else:
            mixed.append(w)
    assert any(m.kept_state is not None for m in mixed) and any(m.kept_state is None for m in mixed)
    ref = np.asarray([helpers.oracle_expectation(w, v, op) for w, v in zip(whole, values)])
    base = np.asarray(evaluator.evaluate_circuits(mixed, values))
    assert np.abs(base - ref).max() < EXP_TOL
    for _ in range(30):
        assert np.array_equal(np.asarray(evaluator.evaluate_circuits(mixed, values)), base)
    # reversed, one at a time, the kept ones alone
    assert np.array_equal(np.asarray(evaluator.evaluate_circuits(mixed[::-1], values[::-1]))[::-1], base)
    for i in (0, 5, 17, 23):
        assert evaluator.evaluate_circuits([mixed[i]], [values[i]])[0] == base[i]
    only = [i for i, m in enumerate(mixed) if m.kept_state is not None]
    assert np.array_equal(np.asarray(evaluator.evaluate_circuits([mixed[i] for i in only], [values[i] for i in only])), base[only])
    # one stream; pushes of five (launch groups cut across the three kinds)
    dev.set_option("streams", 1)
    assert np.array_equal(np.asarray(evaluator.evaluate_circuits(mixed, values)), base)
    dev.set_option("streams", 2)
    dev._push_evals = 5
    try:
        assert np.array_equal(np.asarray(evaluator.evaluate_circuits(mixed, values)), base)
    finally:
        dev._push_evals = 0
    # parameter values resident in device memory, results left on the device
    width = max(len(v) for v in values)
    matrix = np.zeros((len(values), width))
    for i, v in enumerate(values):
        matrix[i, : len(v)] = v
    tensor = torch.from_numpy(matrix).cuda()
    torch.cuda.synchronize()
    assert np.array_equal(np.asarray(evaluator.evaluate_circuits(mixed, tensor)), base)
    assert np.array_equal(np.asarray(evaluator.evaluate_circuits(mixed, tensor)), base)  # (the repeated batch keeps its layout)
    # all kept / all whole give the oracle's values too (not the same bits as each other: another order of operations)
    assert np.abs(np.asarray(evaluator.evaluate_circuits(kept, values)) - ref).max() < EXP_TOL


def test_kept_states_at_twenty_qubits_against_the_c_oracle():
    n, layers = 20, 8
    population = EVQEPopulation.random_population(n, layers, 64, True, 0)
    op = helpers.random_ising_operator(n, seed=2020)
    evaluator = OperatorCircuitEvaluator(op)
    cases = search_circuits(population, layers - 1)
    states = evaluator.keep_states([c[1] for c in cases], [[] for _ in cases])
    kept = [c[2].continue_from(s) for c, s in zip(cases, states)]
    rng = np.random.default_rng(1)
    points = [list(rng.uniform(0, 2 * np.pi, len(c[3]))) for c in cases]
    got = np.asarray(evaluator.evaluate_circuits(kept, points))
    whole = np.asarray(evaluator.evaluate_circuits([c[0] for c in cases], points))
    assert np.abs(got - whole).max() < EXP_TOL
    orc = helpers.load_c_oracle()
    table = orc.diagonal_table(op)
    scratch = np.zeros(2 << n)
    for i in (0, 21, 42, 63):
        assert abs(got[i] - orc.evaluate(cases[i][0], points[i], op, table, scratch)) < EXP_TOL
    costs = evaluator.circuit_costs(kept)
    assert all(c["on_kept_state"] and c["route"] == "gate passes" for c in costs)
    assert np.mean([c["n_passes"] for c in costs]) < np.mean([c["n_passes"] for c in evaluator.circuit_costs([c[0] for c in cases])])


def test_kept_states_in_single_precision():
    n, layers = 14, 6
    population = EVQEPopulation.random_population(n, layers, 8, True, 2)
    op = helpers.random_ising_operator(n, seed=3)
    scale = float(np.abs(op.coeffs).sum())
    ev32 = OperatorCircuitEvaluator(op, dtype="fp32")
    cases = search_circuits(population, layers - 1)
    states = ev32.keep_states([c[1] for c in cases], [[] for _ in cases])
    kept = [c[2].continue_from(s) for c, s in zip(cases, states)]
    got = np.asarray(ev32.evaluate_circuits(kept, [c[3] for c in cases]))
    ref = np.asarray([helpers.oracle_expectation(c[0], c[3], op) for c in cases])
    assert np.abs(got - ref).max() < 2e-6 * scale


def test_kept_state_lifetime_and_refusals():
    n = 13
    population = EVQEPopulation.random_population(n, 3, 4, True, 1)
    op = helpers.random_ising_operator(n, seed=1)
    evaluator = OperatorCircuitEvaluator(op)
    dev = evaluator.statevector_device
    other = OperatorCircuitEvaluator(op)
    cases = search_circuits(population, 2)
    states = evaluator.keep_states([c[1] for c in cases], [[] for _ in cases])
    assert dev.kept_state_count() == 4
    kept = [c[2].continue_from(s) for c, s in zip(cases, states)]
    values = [c[3] for c in cases]
    before = evaluator.evaluate_circuits(kept, values)
    # a state of another device, a pickled circuit, a sampled circuit, a released state: refused
    with pytest.raises(ValueError):
        other.evaluate_circuits([kept[0]], [values[0]])
    with pytest.raises(TypeError):
        pickle.dumps(kept[0])
    with pytest.raises(Exception):
        OperatorSamplerCircuitEvaluator(64, op, statevector_device=dev, seed=1).evaluate_circuits([kept[0]], [values[0]])
    with pytest.raises(ValueError):
        evaluator.keep_states([kept[0]], [values[0]])
    # the KeptState objects go first: the circuits registered on them keep them alive, and keep working
    del states
    gc.collect()
    assert dev.kept_state_count() == 4
    assert evaluator.evaluate_circuits(kept, values) == before
    # ... and new states do not land on theirs
    more = evaluator.keep_states([c[1] for c in cases[:2]], [[], []])
    assert dev.kept_state_count() == 6
    assert evaluator.evaluate_circuits(kept, values) == before
    del kept, more, cases  # (cases holds the continued circuits too: continue_from returns the circuit itself)
    evaluator.forget_circuits()  # (the device remembers the circuit objects of its last call)
    gc.collect()
    assert dev.kept_state_count() == 0
    cases = search_circuits(population, 2)
    # the memory is reused: many rounds of keep / continue / drop do not grow the count
    for _ in range(5):
        states = evaluator.keep_states([c[1] for c in cases], [[] for _ in cases])
        kept = [c[2].continue_from(s) for c, s in zip(search_circuits(population, 2), states)]
        assert evaluator.evaluate_circuits(kept, values) == before
        del states, kept
        evaluator.forget_circuits()
        gc.collect()
    assert dev.kept_state_count() == 0
    # raw C ABI: unknown ids
    lib, handle = dev._lib, dev._handle
    bad = C.c_int(12345)
    assert lib.qsv_prefix_destroy(handle, 1, C.byref(bad)) == _lib.QSV_E_ARG
    out = C.c_int(0)
    ops = cases[0][2].packed()
    assert lib.qsv_circuit_create_on_prefix(handle, 12345, len(ops), _lib.as_ptr(ops), cases[0][2].num_parameters, C.byref(out)) == _lib.QSV_E_ARG


def test_an_initial_state_circuit_is_part_of_the_kept_state():
    from queasars_amd.ir import CircuitIR

    n = 13
    population = EVQEPopulation.random_population(n, 4, 4, True, 4)
    op = helpers.random_ising_operator(n, seed=4)
    initial = CircuitIR(n)
    for q in range(n):
        initial.u(0.3 + 0.1 * q, 0.2, 0.1 * q, q)
    initial.cu3(0.4, 0.1, 0.9, 0, n - 1)
    evaluator = OperatorCircuitEvaluator(op, initial_state_circuit=initial)
    cases = search_circuits(population, 3)
    states = evaluator.keep_states([c[1] for c in cases], [[] for _ in cases])
    kept = [c[2].continue_from(s) for c, s in zip(cases, states)]
    got = np.asarray(evaluator.evaluate_circuits(kept, [c[3] for c in cases]))
    ref = np.asarray([helpers.oracle_expectation(initial.compose(c[0]), c[3], op) for c in cases])
    assert np.abs(got - ref).max() < EXP_TOL


def test_layer_searches_on_kept_states_are_the_searches_on_whole_circuits(monkeypatch):
    """The EVQE driver's layer search with kept states (solver._kept_state_circuits) against the same search on whole circuits:
    the same number of evaluations, iterates within 1e-8 (the values agree to 1e-14, not bit for bit), host-packed and
    device-resident."""
    from queasars_amd.evqe.solver import SPSA, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration

    n, layers = 16, 7
    population = EVQEPopulation.random_population(n, layers, 24, True, 11)
    op = helpers.random_ising_operator(n, seed=15)
    results = {}
    n_kept = 0
    for on_device in (False, True):
        for kept in (False, True):
            monkeypatch.setenv("QSV_KEPT_STATES", "1" if kept else "0")
            evaluator = OperatorCircuitEvaluator(op)
            cfg = EVQEMinimumEigensolverConfiguration(
                optimizer=SPSA(maxiter=8, learning_rate=0.2, perturbation=0.1), population_size=24, max_generations=1, random_seed=0,
                n_initial_layers=layers, randomize_initial_population_parameters=True, speciation_genetic_distance_threshold=2,
                use_tournament_selection=True, tournament_size=2, selection_alpha_penalty=0.1, selection_beta_penalty=0.1,
                parameter_search_probability=0.3, topological_search_probability=0.4, layer_removal_probability=0.05,
                device_resident_search=on_device)
            solver = EVQEMinimumEigensolver(cfg)
            if kept:
                plan = solver._kept_state_circuits(evaluator, list(population.individuals), [-1] * 24)
                assert 0 < len(plan) < 24, len(plan)  # (unsplittable individuals whose rest is fewer passes, and only they)
                n_kept = len(plan)
                del plan
            new, nfev = solver._optimize_layers(evaluator, list(population.individuals), [-1] * 24, list(range(100, 124)))
            results[(on_device, kept)] = (np.concatenate([np.asarray(ind.parameter_values) for ind in new]), nfev)
            gc.collect()
            assert evaluator.statevector_device.kept_state_count() == 0
    for on_device in (False, True):
        whole, with_kept = results[(on_device, False)], results[(on_device, True)]
        assert whole[1] == with_kept[1]
        assert np.abs(whole[0] - with_kept[0]).max() < 1e-8
    assert n_kept >= 4, n_kept


def test_a_single_precision_search_evaluator_under_a_double_precision_fitness(monkeypatch):
    """``compute_minimum_eigenvalue(evaluator, search_evaluator)``: the layer searches -- kept states included -- on an fp32
    handle, the generation's fitness on the fp64 one: the reported eigenvalue is the fp64 handle's value of the best
    individual (the oracle's to 1e-10), and the run lowers the energy as the all-fp64 run does."""
    from queasars_amd.evqe.solver import SPSA, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration

    n = 14
    op = helpers.random_ising_operator(n, seed=21)
    monkeypatch.setenv("QSV_KEPT_STATES", "1")

    def run(search_dtype):
        cfg = EVQEMinimumEigensolverConfiguration(
            optimizer=SPSA(maxiter=10, learning_rate=0.2, perturbation=0.1), population_size=16, max_generations=3, random_seed=3,
            n_initial_layers=5, randomize_initial_population_parameters=True, speciation_genetic_distance_threshold=2,
            use_tournament_selection=True, tournament_size=2, selection_alpha_penalty=0.1, selection_beta_penalty=0.1,
            parameter_search_probability=0.3, topological_search_probability=0.4, layer_removal_probability=0.05)
        fitness = OperatorCircuitEvaluator(op)
        search = OperatorCircuitEvaluator(op, dtype=search_dtype) if search_dtype else None
        return EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(fitness, search)

    mixed, double = run("fp32"), run(None)
    best = mixed.best_individual
    exact = helpers.oracle_expectation(best.get_parameterized_quantum_circuit(), list(best.parameter_values), op)
    assert abs(mixed.eigenvalue - exact) < 1e-10
    assert mixed.generations == double.generations == 3
    assert mixed.best_expectation_values[-1] < mixed.best_expectation_values[0] + 1e-9
    # the two runs see values that differ by 1e-6 and may part ways; they end in the same neighbourhood
    assert abs(mixed.eigenvalue - double.eigenvalue) < 0.25 * abs(double.eigenvalue)
