"""The oracle against itself and against analytic known answers.  Parity is otherwise unpinned:
Qiskit is not installable here and the reference's tests hold no numeric vectors for this path."""

import json
from pathlib import Path

import numpy as np
import pytest

import helpers
from oracle import statevector_oracle as so
from queasars_amd.ir import CircuitIR, PauliOperator

GOLDEN = Path(__file__).parent / "golden"


@pytest.mark.parametrize("n_qubits,n_layers", [(1, 2), (2, 3), (4, 3), (6, 2), (8, 3), (10, 2)])
def test_two_formulations_agree(n_qubits, n_layers):
    _, circuits, params = helpers.population_circuits(n_qubits, n_layers, 3, seed=n_qubits)
    op = helpers.random_pauli_operator(n_qubits, 8, seed=5)
    for c, p in zip(circuits, params):
        ops = c.bound_ops(p)
        a, b = so.simulate(n_qubits, ops), so.simulate_dense(n_qubits, ops)
        assert np.abs(a - b).max() < 1e-13
        e_a = so.pauli_expectation(a, op.x_mask.tolist(), op.z_mask.tolist(), op.coeffs.tolist())
        e_b = so.pauli_expectation_dense(b, op.labels, op.coeffs.tolist())
        assert abs(e_a - e_b) < 1e-13


@pytest.mark.parametrize("n_qubits", [3, 9, 14])
def test_c_oracle_matches_numpy_oracle(n_qubits, c_oracle):
    _, circuits, params = helpers.population_circuits(n_qubits, 3, 3, seed=17)
    general = helpers.random_pauli_operator(n_qubits, 10, seed=2)
    ising = helpers.random_ising_operator(n_qubits, seed=3)
    table = c_oracle.diagonal_table(ising)
    assert np.array_equal(table, so.diagonal_values(n_qubits, ising.z_mask.tolist(), ising.coeffs.real.tolist()))
    for c, p in zip(circuits, params):
        ref = helpers.oracle_state(c, p)
        assert np.abs(c_oracle.simulate(c, p) - ref).max() < 1e-13
        for op in (general, ising):
            want = so.pauli_expectation(ref, op.x_mask.tolist(), op.z_mask.tolist(), op.coeffs.tolist()).real
            assert abs(c_oracle.evaluate(c, p, op) - want) < 1e-12
        assert abs(c_oracle.evaluate(c, p, ising, table) - helpers.oracle_expectation(c, p, ising)) < 1e-12


def test_u_matrix_definition():
    # Qiskit UGate: U(pi/2, 0, pi) = H, U(pi, 0, pi) = X, U(pi, pi/2, pi/2) = Y, U(0, 0, lam) = phase gate
    h = so.u_matrix(np.pi / 2, 0.0, np.pi)
    assert np.allclose(h, np.array([[1, 1], [1, -1]]) / np.sqrt(2))
    assert np.allclose(so.u_matrix(np.pi, 0.0, np.pi), [[0, 1], [1, 0]])
    assert np.allclose(so.u_matrix(np.pi, np.pi / 2, np.pi / 2), [[0, -1j], [1j, 0]])
    assert np.allclose(so.u_matrix(0.0, 0.0, 0.7), [[1, 0], [0, np.exp(0.7j)]])
    m = so.u_matrix(0.3, 1.1, -0.4)
    assert np.allclose(m.conj().T @ m, np.eye(2))


def test_little_endian_and_cu3_roles():
    # X on qubit 1 of 3 -> index 2; cu3 acts on the target only when the control is 1
    assert np.argmax(np.abs(so.simulate(3, [(so.U, 1, -1, np.pi, 0.0, np.pi)]))) == 2
    flip_t = (so.CU3, 2, 0, np.pi, 0.0, np.pi)  # control 0, target 2
    assert np.argmax(np.abs(so.simulate(3, [flip_t]))) == 0
    assert np.argmax(np.abs(so.simulate(3, [(so.U, 0, -1, np.pi, 0.0, np.pi), flip_t]))) == 0b101


def test_pauli_label_convention_and_y_phase():
    assert so.label_to_masks("IZ") == (0, 1) and so.label_to_masks("XI") == (2, 0) and so.label_to_masks("YI") == (2, 2)
    plus_i = np.array([1, 1j]) / np.sqrt(2)  # +1 eigenstate of Y
    assert abs(so.pauli_term_expectation(plus_i, 1, 1) - 1.0) < 1e-15
    state = so.simulate(2, [(so.U, 0, -1, 0.4, 0.3, 0.2), (so.CU3, 1, 0, 1.0, -0.5, 0.25)])
    for label in ("XY", "YZ", "ZX", "YY", "IX"):
        x, z = so.label_to_masks(label)
        assert abs(so.pauli_term_expectation(state, x, z) - np.vdot(state, so.dense_pauli(label) @ state)) < 1e-14


def test_zero_angle_identity_known_answer():
    """Reference fixtures start from zero angles (test_evqe_operators.py:36-38): the state stays |0..0>."""
    _, circuits, params = helpers.population_circuits(5, 2, 4, seed=0, randomize=False)
    op = PauliOperator(["ZIIII", "IZZII", "XIIII", "IIYIZ", "IIIII"], [0.5, -1.25, 3.0, 2.0, 0.75])
    for c, p in zip(circuits, params):
        assert abs(helpers.oracle_expectation(c, p, op) - (0.5 - 1.25 + 0.75)) < 1e-15


def test_reference_test_hamiltonian_known_answers():
    """Hand restatement of the reference's `min x^2 - y^2` Ising Hamiltonian."""
    op = PauliOperator.from_sparse_list(
        [("Z", [0], -1.5), ("Z", [1], -3.0), ("ZZ", [0, 1], 1.0), ("Z", [2], 1.5), ("Z", [3], 3.0), ("ZZ", [2, 3], -1.0)], 4
    )
    diag = so.diagonal_values(4, op.z_mask.tolist(), op.coeffs.real.tolist())
    assert diag[0] == 0.0
    for state in range(16):
        x, y = state & 3, state >> 2
        # offset-free Ising form: value differs from x^2 - y^2 by a constant (0 here)
        assert abs(diag[state] - (x * x - y * y)) < 1e-12
    assert diag.min() == -9.0 and int(np.argmin(diag)) == 0b1100


def test_cvar_accumulation():
    items = [("a", 0.25, 4.0), ("b", 0.25, 1.0), ("c", 0.5, 2.0)]
    assert abs(so.cvar_expectation(items, 1.0) - (0.25 * 4 + 0.25 * 1 + 0.5 * 2)) < 1e-15
    assert abs(so.cvar_expectation(items, 0.5) - (0.25 * 1 + 0.25 * 2) / 0.5) < 1e-15
    assert abs(so.cvar_expectation(items, 0.25) - 1.0) < 1e-15
    dist = {0b00: 0.5, 0b01: 0.25, 0b11: 0.25}
    z, c = [1, 2], [1.0, 0.5]
    assert abs(so.expectation_from_distribution(dist, z, c, 1.0) - (0.5 * 1.5 + 0.25 * -0.5 + 0.25 * -1.5)) < 1e-15
    assert abs(so.expectation_from_distribution(dist, z, c, 0.25) - (-1.5)) < 1e-15


def test_golden_fixtures():
    """Committed vectors (tests/golden/make_golden.py): seeds -> ops, final state (small n) and <H>."""
    data = json.loads((GOLDEN / "evqe_small.json").read_text())
    assert len(data["cases"]) >= 12
    for case in data["cases"]:
        n = case["n_qubits"]
        ops = [tuple(op) for op in case["ops"]]
        state = so.simulate(n, ops)
        x = [so.label_to_masks(label)[0] for label in case["labels"]]
        z = [so.label_to_masks(label)[1] for label in case["labels"]]
        got = so.pauli_expectation(state, x, z, case["coeffs"]).real
        assert abs(got - case["expectation"]) < 1e-12
        if "state_re" in case:
            want = np.asarray(case["state_re"]) + 1j * np.asarray(case["state_im"])
            assert np.abs(state - want).max() < 1e-13


def test_textbook_expectation_values():
    """Bell and GHZ correlations with Mermin's signs, a controlled phase on |++>, the Bloch vector of U(theta, phi, .)|0>, a
    controlled rotation behind a set control: values any textbook gives, no simulator involved."""
    for name, circuit, values in helpers.textbook_cases():
        for label, want in values.items():
            got = helpers.oracle_expectation(circuit, [], PauliOperator([label], [1.0]))
            assert abs(got - want) < 1e-14, (name, label, got, want)
