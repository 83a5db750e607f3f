"""GPU parity tests: the HIP path (through the C ABI) against the oracle.  Run on the MI355X box with -m gpu.

Tolerances: fp64 expectation values within 1e-10 of the oracle (BASELINE.json north_star); amplitudes within
1e-12; fp32 states within 2e-5 per amplitude.
"""

import numpy as np
import pytest

import helpers
from oracle import statevector_oracle as so
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, StatevectorDevice
from queasars_amd.ir import CircuitIR, ParamRef, PauliOperator

pytestmark = pytest.mark.gpu

EXP_TOL = 1e-10
AMP_TOL = 1e-12


@pytest.mark.parametrize("n_qubits,n_layers", [(1, 2), (2, 3), (3, 2), (5, 3), (7, 2), (8, 2), (10, 3), (12, 4), (13, 3), (14, 4), (16, 2)])
def test_statevector_matches_oracle(n_qubits, n_layers):
    _, circuits, params = helpers.population_circuits(n_qubits, n_layers, 4, seed=11)
    dev = StatevectorDevice(n_qubits)
    for c, p in zip(circuits, params):
        got = dev.statevector(c, p)
        ref = helpers.oracle_state(c, p)
        assert np.abs(got - ref).max() < AMP_TOL


@pytest.mark.parametrize(
    "cfg",
    [
        dict(tile_bits=10, reg_bits=3, low_bits=3),
        dict(tile_bits=11, reg_bits=4, low_bits=4),
        dict(tile_bits=9, reg_bits=2, low_bits=2),
        dict(tile_bits=12, reg_bits=4, low_bits=2),
        dict(tile_bits=13, reg_bits=4, low_bits=2),
        dict(tile_bits=12, reg_bits=3, exchange=2),
        dict(tile_bits=12, reg_bits=3, exchange=3),
        dict(tile_bits=11, reg_bits=2, exchange=3, group=2),
    ],
)
def test_statevector_other_geometries(cfg):
    n_qubits = 14
    _, circuits, params = helpers.population_circuits(n_qubits, 3, 3, seed=5)
    dev = StatevectorDevice(n_qubits, **cfg)
    for c, p in zip(circuits, params):
        assert np.abs(dev.statevector(c, p) - helpers.oracle_state(c, p)).max() < AMP_TOL


@pytest.mark.parametrize(
    "n_qubits,cfg",
    [
        (13, dict(tile_bits=7, reg_bits=2, low_bits=2)),
        (14, dict(tile_bits=8, reg_bits=2, low_bits=1)),
        (15, dict(tile_bits=9, reg_bits=3, low_bits=3)),
        (16, dict(tile_bits=10, reg_bits=3, low_bits=2)),
        (17, dict(tile_bits=11, reg_bits=3, low_bits=2)),
        (18, dict(tile_bits=12, reg_bits=3, low_bits=1)),
        (16, dict(tile_bits=8, reg_bits=2, low_bits=2, dtype="fp32")),
    ],
)
def test_compact_first_pass_on_device(n_qubits, cfg):
    """Small tiles leave many outer qubits: most of these circuits take the compact first pass (plan.hpp COMPACT) and
    several passes; amplitudes and fused expectation values against the oracle."""
    cfg = dict(cfg)
    dtype = cfg.pop("dtype", "fp64")
    _, circuits, params = helpers.population_circuits(n_qubits, 3, 5, seed=17)
    dev = StatevectorDevice(n_qubits, dtype=dtype, **cfg)
    tol = AMP_TOL if dtype == "fp64" else 2e-5
    for c, p in zip(circuits, params):
        assert np.abs(dev.statevector(c, p) - helpers.oracle_state(c, p)).max() < tol
    op = helpers.random_ising_operator(n_qubits, seed=4)
    got = OperatorCircuitEvaluator(op, statevector_device=dev).evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < (EXP_TOL if dtype == "fp64" else 1e-3)


def test_config1_plumbing_general_paulis():
    """BASELINE config 1: n=8, P=4, L=2, 20 random Pauli strings."""
    n = 8
    _, circuits, params = helpers.population_circuits(n, 2, 4, seed=0)
    op = helpers.random_pauli_operator(n, 20, seed=1234)
    ev = OperatorCircuitEvaluator(op)
    got = ev.evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < EXP_TOL


@pytest.mark.parametrize("n_qubits", [4, 9, 12, 15])
def test_diagonal_expectation_matches_oracle(n_qubits):
    _, circuits, params = helpers.population_circuits(n_qubits, 3, 6, seed=3)
    op = helpers.random_ising_operator(n_qubits, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    got = ev.evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < EXP_TOL


@pytest.mark.parametrize("n_qubits", [3, 10, 13])
def test_general_expectation_matches_oracle(n_qubits):
    _, circuits, params = helpers.population_circuits(n_qubits, 2, 5, seed=9)
    op = helpers.random_pauli_operator(n_qubits, 12, seed=77)
    ev = OperatorCircuitEvaluator(op)
    got = ev.evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < EXP_TOL


def test_general_operator_population_in_several_pushes():
    """A population of 40 goes to the device in three pushes; with X / Y terms the expectation kernels run per launch
    group and must find each group's states where the passes left them."""
    n = 10
    _, circuits, params = helpers.population_circuits(n, 2, 40, seed=9)
    op = helpers.random_pauli_operator(n, 12, seed=31)
    got = OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < EXP_TOL


def test_editing_a_circuit_after_evaluation_is_noticed():
    """Batch metadata is cached by object identity between calls; editing a circuit in place must void it."""
    n = 6
    op = helpers.random_ising_operator(n, seed=1)
    ev = OperatorCircuitEvaluator(op)
    c = CircuitIR(n).u(0.3, 0.1, 0.2, 0).cu3(0.5, 0.2, 0.1, 0, 3)
    circuits = [c, CircuitIR(n).u(1.1, 0.0, 0.0, 2)]
    first = ev.evaluate_circuits(circuits, [[], []])
    assert abs(first[0] - helpers.oracle_expectation(c, [], op)) < EXP_TOL
    c.u(0.9, 0.4, 0.0, 5)  # same object, one more gate
    second = ev.evaluate_circuits(circuits, [[], []])
    assert abs(second[0] - helpers.oracle_expectation(c, [], op)) < EXP_TOL and second[0] != first[0]
    assert second[1] == first[1]


def test_concurrent_callers_share_one_evaluator():
    """The reference calls evaluate_circuits from up to population_size threads at once (evqe.py:232-236,
    selection.py:75-85, one circuit per call): calls on one handle are serialised and every caller gets its own result."""
    from concurrent.futures import ThreadPoolExecutor

    n = 11
    _, circuits, params = helpers.population_circuits(n, 2, 16, seed=12)
    ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=6))
    serial = [ev.evaluate_circuits([c], [p])[0] for c, p in zip(circuits, params)]
    with ThreadPoolExecutor(max_workers=16) as pool:
        for _ in range(3):
            futures = [pool.submit(ev.evaluate_circuits, [c], [p]) for c, p in zip(circuits, params)]
            assert [f.result()[0] for f in futures] == serial


def test_primitive_shaped_front_ends():
    """queasars_amd.primitives: the two call shapes the reference's evaluators use on Qiskit primitives
    (circuit_evaluation.py:204-215, :50-59), here with CircuitIR pubs."""
    from queasars_amd.primitives import GpuEstimator, GpuSampler

    n = 9
    _, circuits, params = helpers.population_circuits(n, 2, 5, seed=3)
    op_a, op_b = helpers.random_ising_operator(n, seed=1), helpers.random_pauli_operator(n, 6, seed=2)
    pubs = [(c, op_a if i % 2 == 0 else op_b, p) for i, (c, p) in enumerate(zip(circuits, params))]
    results = GpuEstimator().run(pubs=pubs, precision=0.0).result()
    got = [float(np.real(r.data.evs)) for r in results]
    ref = [helpers.oracle_expectation(c, p, op) for c, op, p in pubs]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < EXP_TOL
    noisy = [float(r.data.evs) for r in GpuEstimator(seed=5).run(pubs=pubs, precision=0.05).result()]
    assert 0 < np.abs(np.asarray(noisy) - np.asarray(ref)).max() < 0.5
    # sampler: a basis state is measured with certainty; bitstrings have the highest qubit leftmost
    flip = CircuitIR(n).u(np.pi, 0.0, np.pi, 2).u(np.pi, 0.0, np.pi, 7)
    counts = GpuSampler(n, seed=3).run(pubs=[(flip, [])], shots=200).result()[0].data["meas"].get_counts()
    assert counts == {format((1 << 2) | (1 << 7), f"0{n}b"): 200}
    shots = 4000
    res = GpuSampler(n, seed=9).run(pubs=[(circuits[0], params[0])], shots=shots).result()[0].data["meas"].get_int_counts()
    probs = np.abs(helpers.oracle_state(circuits[0], params[0])) ** 2
    top = int(np.argmax(probs))
    assert abs(res.get(top, 0) / shots - probs[top]) < 6 * np.sqrt(probs[top] * (1 - probs[top]) / shots) + 2 / shots


def test_evaluators_from_configured_primitives():
    """configured_primitives.evaluator_for: the three evaluators a solver configured the reference's way gets
    (evolving_ansatz_minimum_eigensolver.py builds them from ConfiguredEstimatorV2 / ConfiguredSamplerV2)."""
    from queasars_amd.circuit_evaluation import (
        BitstringCircuitEvaluator,
        BitstringEvaluator,
        ConfiguredEstimatorV2,
        ConfiguredSamplerV2,
        OperatorSamplerCircuitEvaluator,
        evaluator_for,
    )
    from queasars_amd.primitives import GpuEstimator, GpuSampler

    n = 8
    _, circuits, params = helpers.population_circuits(n, 2, 4, seed=21)
    op = helpers.random_ising_operator(n, seed=4)
    exact = evaluator_for(ConfiguredEstimatorV2(GpuEstimator(), 0.0), operator=op)
    assert isinstance(exact, OperatorCircuitEvaluator)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(exact.evaluate_circuits(circuits, params)) - np.asarray(ref)).max() < EXP_TOL
    configured = ConfiguredSamplerV2(GpuSampler(n, seed=11), 20000)
    sampled = evaluator_for(configured, operator=op)
    assert isinstance(sampled, OperatorSamplerCircuitEvaluator)
    spread = float(np.abs(op.coeffs).sum())
    got = sampled.evaluate_circuits(circuits, params)
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < 6 * spread / np.sqrt(20000)
    ones = evaluator_for(configured, bitstring_evaluator=BitstringEvaluator(n, lambda b: float(b.count("1"))))
    assert isinstance(ones, BitstringCircuitEvaluator)
    want = [
        float(np.dot(np.abs(helpers.oracle_state(c, p)) ** 2, [bin(i).count("1") for i in range(1 << n)]))
        for c, p in zip(circuits, params)
    ]
    assert np.abs(np.asarray(ones.evaluate_circuits(circuits, params)) - np.asarray(want)).max() < 6 * n / np.sqrt(20000)


def test_zero_angles_known_answer():
    """u(0,0,0) = cu3(0,0,0) = I: the state stays |0..0>, <H> = sum of the I/Z-only coefficients."""
    n = 6
    _, circuits, params = helpers.population_circuits(n, 2, 3, seed=0, randomize=False)
    op = PauliOperator(["ZIIIII", "IZZIII", "XIIIII", "IIYIIZ", "IIIIII"], [0.5, -1.25, 3.0, 2.0, 0.75])
    ev = OperatorCircuitEvaluator(op)
    got = ev.evaluate_circuits(circuits, params)
    assert np.allclose(got, 0.5 - 1.25 + 0.75, atol=1e-14)


def test_bit_flip_known_answer():
    """u(pi, 0, pi) = X on qubit q gives basis state 1 << q."""
    n = 9
    dev = StatevectorDevice(n)
    for q in range(n):
        c = CircuitIR(n).u(np.pi, 0.0, np.pi, q)
        probs = dev.probabilities(c, [])
        assert abs(probs[1 << q] - 1.0) < 1e-14 and abs(probs.sum() - 1.0) < 1e-14


def test_reference_test_hamiltonian_known_answer():
    """min x^2 - y^2 Ising form: <0000|H|0000> = 0 and the minimum -9 sits at x=0, y=3."""
    op = PauliOperator.from_sparse_list(
        [("Z", [0], -1.5), ("Z", [1], -3.0), ("ZZ", [0, 1], 1.0), ("Z", [2], 1.5), ("Z", [3], 3.0), ("ZZ", [2, 3], -1.0)], 4
    )
    ev = OperatorCircuitEvaluator(op)
    ident = CircuitIR(4).id(0)
    assert abs(ev.evaluate_circuits([ident], [[]])[0]) < 1e-14
    flip_y = CircuitIR(4).u(np.pi, 0.0, np.pi, 2).u(np.pi, 0.0, np.pi, 3)  # y = 3, x = 0
    assert abs(ev.evaluate_circuits([flip_y], [[]])[0] - (-9.0)) < 1e-13


def test_partially_parameterized_and_param_indices():
    n = 10
    pop, _, _ = helpers.population_circuits(n, 3, 3, seed=21)
    op = helpers.random_ising_operator(n, seed=1)
    ev = OperatorCircuitEvaluator(op)
    for ind in pop.individuals:
        full = ind.get_parameterized_quantum_circuit()
        ref = helpers.oracle_expectation(full, list(ind.parameter_values), op)
        for layer in range(len(ind.layers)):
            part = ind.get_partially_parameterized_quantum_circuit({layer})
            vals = list(ind.get_layer_parameter_values(layer))
            got = ev.evaluate_circuits([part, part], [vals, vals])
            assert abs(got[0] - ref) < EXP_TOL and got[0] == got[1]


def test_results_ordered_by_input_index_and_batch_invariance():
    n = 11
    _, circuits, params = helpers.population_circuits(n, 2, 9, seed=4)
    op = helpers.random_ising_operator(n, seed=8)
    ev = OperatorCircuitEvaluator(op)
    together = ev.evaluate_circuits(circuits, params)
    one_by_one = [ev.evaluate_circuits([c], [p])[0] for c, p in zip(circuits, params)]
    assert together == one_by_one  # bitwise: fixed-order reductions
    rev = ev.evaluate_circuits(circuits[::-1], params[::-1])
    assert rev[::-1] == together


def test_two_streams_and_push_plans_do_not_change_results(monkeypatch):
    """A population goes to the device in several pushes on two HIP streams; one push on one stream must give the
    same bits (every evaluation has its own state slot and its own partial sums)."""
    n = 14
    _, circuits, params = helpers.population_circuits(n, 3, 40, seed=21)
    op = helpers.random_ising_operator(n, seed=5)
    default = OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits[:4], params[:4])]
    assert np.abs(np.asarray(default[:4]) - np.asarray(ref)).max() < EXP_TOL
    monkeypatch.setenv("QSV_STREAMS", "1")
    monkeypatch.setenv("QSV_PUSH_EVALS", "40")
    assert OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params) == default
    monkeypatch.setenv("QSV_STREAMS", "2")
    monkeypatch.setenv("QSV_PUSH_EVALS", "7")
    assert OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params) == default


def test_evaluator_survives_pickling():
    """Process-based executors (the reference supports Dask workers, evqe.py:39-44) rebuild the evaluator from plain
    data on the other side: the clone owns a new device handle and gives the same bits."""
    import pickle

    n = 9
    _, circuits, params = helpers.population_circuits(n, 2, 3, seed=6)
    ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2))
    want = ev.evaluate_circuits(circuits, params)
    clone = pickle.loads(pickle.dumps(ev))
    moved = pickle.loads(pickle.dumps(circuits))
    assert clone.n_qubits == n and clone.evaluate_circuits(moved, params) == want


def test_initial_state_circuit():
    n = 5
    _, circuits, params = helpers.population_circuits(n, 2, 2, seed=13)
    op = helpers.random_pauli_operator(n, 6, seed=3)
    init = CircuitIR(n).u(0.3, 0.1, -0.2, 0).cu3(1.0, 0.2, 0.3, 0, 3)
    ev = OperatorCircuitEvaluator(op, initial_state_circuit=init)
    got = ev.evaluate_circuits(circuits, params)
    for g, c, p in zip(got, circuits, params):
        ops = init.bound_ops([]) + c.bound_ops(p)
        state = so.simulate(n, ops)
        ref = so.pauli_expectation(state, op.x_mask.tolist(), op.z_mask.tolist(), op.coeffs.tolist()).real
        assert abs(g - ref) < EXP_TOL


def test_fp32_state_close_to_fp64():
    n = 12
    _, circuits, params = helpers.population_circuits(n, 3, 2, seed=2)
    dev = StatevectorDevice(n, dtype="fp32")
    for c, p in zip(circuits, params):
        assert np.abs(dev.statevector(c, p) - helpers.oracle_state(c, p)).max() < 2e-5


def test_random_populations_against_the_c_oracle(c_oracle):
    """A sweep over sizes, depths and seeds (default geometry: compact first pass whenever n > 13), expectation values
    against the plain-C oracle."""
    rng = np.random.default_rng(77)
    worst = 0.0
    for trial in range(24):
        n = int(rng.integers(8, 20))
        layers = int(rng.integers(1, 7))
        _, circuits, params = helpers.population_circuits(n, layers, 3, seed=int(rng.integers(0, 10**6)))
        op = helpers.random_ising_operator(n, seed=trial)
        got = OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params)
        ref = [c_oracle.evaluate(c, p, op) for c, p in zip(circuits, params)]
        worst = max(worst, float(np.abs(np.asarray(got) - np.asarray(ref)).max()))
    assert worst < EXP_TOL


# ---- size-independent properties at the benchmark sizes --------------------------------------------------


@pytest.mark.parametrize("n_qubits", [20, 24])
def test_round_trip_and_norm_at_full_size(n_qubits, c_oracle):
    """circuit followed by its inverse returns |0..0>; <I> = 1; expectation is linear in the coefficients."""
    _, circuits, params = helpers.population_circuits(n_qubits, 4, 2, seed=0)
    ident = PauliOperator(["I" * n_qubits], [1.0])
    dev = StatevectorDevice(n_qubits)
    ev_id = OperatorCircuitEvaluator(ident, statevector_device=dev)
    assert np.allclose(ev_id.evaluate_circuits(circuits, params), 1.0, atol=1e-12)
    c, p = circuits[0], params[0]
    round_trip = helpers.bound_copy(c, p).compose(helpers.inverse_circuit(c, p))
    probs = dev.probabilities(round_trip, [])
    assert abs(probs[0] - 1.0) < 1e-11 and abs(probs.sum() - 1.0) < 1e-11
    # linearity: <a H1 + b H2> = a <H1> + b <H2>
    h1 = helpers.random_ising_operator(n_qubits, seed=1)
    h2 = helpers.random_ising_operator(n_qubits, seed=2)
    combo = PauliOperator(h1.labels + h2.labels, np.concatenate([0.3 * h1.coeffs, -1.7 * h2.coeffs]))
    e1 = OperatorCircuitEvaluator(h1, statevector_device=dev).evaluate_circuits(circuits, params)
    e2 = OperatorCircuitEvaluator(h2, statevector_device=dev).evaluate_circuits(circuits, params)
    e12 = OperatorCircuitEvaluator(combo, statevector_device=dev).evaluate_circuits(circuits, params)
    assert np.abs(np.asarray(e12) - (0.3 * np.asarray(e1) - 1.7 * np.asarray(e2))).max() < 1e-9
    if n_qubits == 20:
        # direct check against the C oracle at the benchmark size
        ref = [c_oracle.evaluate(ci, pi, h1) for ci, pi in zip(circuits, params)]
        assert np.abs(np.asarray(e1) - np.asarray(ref)).max() < EXP_TOL


def test_states_beyond_32_bit_byte_offsets():
    """n = 29: a state is 8 GiB, byte offsets inside it no longer fit 32 bits and the pass kernel takes its 64-bit
    addressing path.  Checked through properties: <I> = 1, <Z_q> of a circuit followed by its inverse is 1 for
    every q (diagonal operator, fused reduction), and a general operator on the highest qubits (pair kernels)."""
    n = 29
    _, circuits, params = helpers.population_circuits(n, 2, 1, seed=3)
    c, p = circuits[0], params[0]
    dev = StatevectorDevice(n)
    round_trip = helpers.bound_copy(c, p).compose(helpers.inverse_circuit(c, p))
    weights = np.linspace(0.5, 1.5, n)
    z_sum = PauliOperator.from_sparse_list([("Z", [q], float(w)) for q, w in enumerate(weights)], n)
    ev = OperatorCircuitEvaluator(z_sum, statevector_device=dev)
    assert abs(ev.evaluate_circuits([round_trip], [[]])[0] - weights.sum()) < 1e-9
    ident = PauliOperator(["I" * n], [1.0])
    assert abs(OperatorCircuitEvaluator(ident, statevector_device=dev).evaluate_circuits([c], [p])[0] - 1.0) < 1e-11
    # one u gate on the top qubit: <X> = sin(theta) cos(phi), <Y> = sin(theta) sin(phi), <Z> = cos(theta)
    theta, phi = 0.7, 0.4
    single = CircuitIR(n).u(theta, phi, 0.3, n - 1).u(0.2, 0.1, 0.0, 0)
    xyz = PauliOperator.from_sparse_list([("X", [n - 1], 1.0), ("Y", [n - 1], 10.0), ("Z", [n - 1], 100.0)], n)
    want = np.sin(theta) * np.cos(phi) + 10.0 * np.sin(theta) * np.sin(phi) + 100.0 * np.cos(theta)
    assert abs(OperatorCircuitEvaluator(xyz, statevector_device=dev).evaluate_circuits([single], [[]])[0] - want) < 1e-9


# ---- sampler branch --------------------------------------------------------------------------------------


def test_sampling_is_seeded_and_follows_the_distribution():
    n = 10
    _, circuits, params = helpers.population_circuits(n, 3, 1, seed=6)
    dev = StatevectorDevice(n)
    probs = dev.probabilities(circuits[0], params[0])
    assert np.abs(probs - so.probabilities(helpers.oracle_state(circuits[0], params[0]))).max() < 1e-13
    shots = 200_000
    a = dev.sample(circuits[0], params[0], shots, seed=123)
    b = dev.sample(circuits[0], params[0], shots, seed=123)
    c = dev.sample(circuits[0], params[0], shots, seed=124)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.max() < (1 << n)
    freq = np.bincount(a.astype(np.int64), minlength=1 << n) / shots
    # every bin within 6 sigma of its binomial expectation (plus two counts: rare bins are Poisson, not Gaussian)
    sigma = np.sqrt(probs * (1 - probs) / shots)
    assert np.all(np.abs(freq - probs) < 6 * sigma + 2.0 / shots)
    assert np.all(freq[probs == 0.0] == 0.0)


def test_basis_state_is_sampled_exactly():
    n = 13
    dev = StatevectorDevice(n)
    c = CircuitIR(n).u(np.pi, 0.0, np.pi, 3).u(np.pi, 0.0, np.pi, 12)
    assert set(dev.sample(c, [], 1000, seed=5).tolist()) == {(1 << 3) | (1 << 12)}


def test_sampler_and_bitstring_evaluators_converge_to_exact_values():
    from queasars_amd.circuit_evaluation import BitstringCircuitEvaluator, BitstringEvaluator, OperatorSamplerCircuitEvaluator

    n = 8
    _, circuits, params = helpers.population_circuits(n, 2, 3, seed=8)
    op = helpers.random_ising_operator(n, seed=4)
    exact = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    shots = 400_000
    got = OperatorSamplerCircuitEvaluator(shots, op, alpha=1.0, seed=1).evaluate_circuits(circuits, params)
    spread = float(np.abs(op.coeffs.real).sum())
    assert np.abs(np.asarray(got) - np.asarray(exact)).max() < 6 * spread / np.sqrt(shots)
    # CVaR(alpha) from samples against CVaR of the exact distribution
    alpha = 0.25
    cvar = OperatorSamplerCircuitEvaluator(shots, op, alpha=alpha, seed=2).evaluate_circuits(circuits, params)
    for value, c, p in zip(cvar, circuits, params):
        probs = so.probabilities(helpers.oracle_state(c, p))
        dist = {i: float(pr) for i, pr in enumerate(probs) if pr > 0}
        want = so.expectation_from_distribution(dist, op.z_mask.tolist(), op.coeffs.tolist(), alpha)
        assert abs(value - want) < 0.05 * spread
    # bitstring evaluator: number of ones in the measured string = sum_q (1 - <Z_q>) / 2
    ones = BitstringEvaluator(n, lambda b: float(b.count("1")))
    got = BitstringCircuitEvaluator(shots, ones, seed=3).evaluate_circuits(circuits, params)
    for value, c, p in zip(got, circuits, params):
        probs = so.probabilities(helpers.oracle_state(c, p))
        want = sum(pr * bin(i).count("1") for i, pr in enumerate(probs))
        assert abs(value - want) < 6 * n / np.sqrt(shots)


# ---- JSSP operator (BASELINE config 4) ---------------------------------------------------------------------


def test_jssp_notebook_energy_on_gpu():
    """The notebook's 12-qubit JSSP Hamiltonian: the basis state of its best schedule has energy 22.75
    (examples/evqe_jssp_optimization.ipynb:384-392), through the estimator and the sampler/CVaR branch."""
    import jssp_instances as inst
    from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator
    from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

    enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
    op = enc.get_problem_hamiltonian()
    starts = {}
    for job, times in zip(enc.jssp_instance.jobs, [(0, 1, 2), (1, 3, 4)]):
        starts.update(dict(zip(job.operations, times)))
    bitstring = enc.bitstring_of(starts)
    prep = CircuitIR(enc.n_qubits)
    for q, bit in enumerate(bitstring[::-1]):
        if bit == "1":
            prep.u(np.pi, 0.0, np.pi, q)
        else:
            prep.id(q)
    assert abs(OperatorCircuitEvaluator(op).evaluate_circuits([prep], [[]])[0] - 22.75) < 1e-10
    sampled = OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=1).evaluate_circuits([prep], [[]])[0]
    assert abs(sampled - 22.75) < 1e-9
    # a random population agrees with the oracle on this operator too
    _, circuits, params = helpers.population_circuits(enc.n_qubits, 2, 4, seed=3)
    got = OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < 1e-9


def test_runtime_notebook_energy_on_gpu():
    """examples/using_the_ibm_runtime.ipynb: the 8-qubit "Simple Instance" (cell 2, encoder arguments of cell 6) has minimum
    energy 22.75 (cell 8's output), at the makespan-4 schedule cell 14 prints -- through the estimator, the sampler / CVaR
    branch and the exact-probability CVaR."""
    import jssp_instances as inst
    from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator
    from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

    enc = JSSPDomainWallHamiltonianEncoder(inst.runtime_simple_instance(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 8
    op = enc.get_problem_hamiltonian()
    starts = {}
    for job, times in zip(enc.jssp_instance.jobs, [(1, 3), (0, 1)]):
        starts.update(dict(zip(job.operations, times)))
    bitstring = enc.bitstring_of(starts)
    prep = CircuitIR(enc.n_qubits)
    for q, bit in enumerate(bitstring[::-1]):
        if bit == "1":
            prep.u(np.pi, 0.0, np.pi, q)
        else:
            prep.id(q)
    assert abs(OperatorCircuitEvaluator(op).evaluate_circuits([prep], [[]])[0] - 22.75) < 1e-10
    assert abs(OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=1).evaluate_circuits([prep], [[]])[0] - 22.75) < 1e-9
    assert abs(OperatorSamplerCircuitEvaluator(None, op, alpha=0.5).evaluate_circuits([prep], [[]])[0] - 22.75) < 1e-10


def test_config4_jssp_end_to_end_evqe():
    """BASELINE config 4: EVQE (sampler + CVaR 0.5, 512 shots, SPSA 33 iterations, population 10) on the notebook's
    12-qubit JSSP instance reaches the notebook's final energy 22.75 with a valid makespan-5 schedule
    (examples/evqe_jssp_optimization.ipynb:384-392, :481)."""
    import jssp_instances as inst
    from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator
    from queasars_amd.evqe.solver import (
        SPSA, BestIndividualRelativeChangeTolerance, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration, SPSATerminationChecker,
    )
    from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

    enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
    evaluator = OperatorSamplerCircuitEvaluator(512, enc.get_problem_hamiltonian(), alpha=0.5, seed=0)
    cfg = EVQEMinimumEigensolverConfiguration(
        optimizer=SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True,
                       termination_checker=SPSATerminationChecker(0.01, 2)),
        population_size=10, max_generations=8, termination_criterion=BestIndividualRelativeChangeTolerance(0.01, 1),
        random_seed=0, n_initial_layers=2, randomize_initial_population_parameters=True,
        speciation_genetic_distance_threshold=1, use_tournament_selection=True, tournament_size=2,
        selection_alpha_penalty=0.15, selection_beta_penalty=0.02, parameter_search_probability=0.39,
        topological_search_probability=0.79, layer_removal_probability=0.02,
    )
    result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(evaluator)
    assert abs(result.eigenvalue - 22.75) < 1e-6
    best = result.best_individual
    probs = evaluator.statevector_device.probabilities(best.get_parameterized_quantum_circuit(), list(best.parameter_values))
    schedule = enc.translate_result_bitstring(format(int(np.argmax(probs)), f"0{enc.n_qubits}b"))
    assert schedule.is_valid and schedule.makespan == 5


def test_textbook_expectation_values_on_the_device():
    """The same textbook facts the oracle is held to (tests/helpers.py textbook_cases: Bell and GHZ correlations with Mermin's
    signs, a controlled phase, the Bloch vector, a controlled rotation behind a set control) through the device path, one
    Pauli string at a time and all of a case's strings as one operator."""
    for name, circuit, values in helpers.textbook_cases():
        for label, want in values.items():
            got = OperatorCircuitEvaluator(PauliOperator([label], [1.0])).evaluate_circuits([circuit], [[]])[0]
            assert abs(got - want) < 1e-12, (name, label, got, want)
        labels = list(values)
        coeffs = [0.5 + 0.25 * i for i in range(len(labels))]
        got = OperatorCircuitEvaluator(PauliOperator(labels, coeffs)).evaluate_circuits([circuit], [[]])[0]
        assert abs(got - sum(c * values[l] for c, l in zip(coeffs, labels))) < 1e-12, name
