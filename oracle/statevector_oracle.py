"""NumPy complex128 oracle for the circuit-evaluation hot path (TEST INFRASTRUCTURE, parity unpinned).

See ``oracle/__init__.py`` for the rules on who may import this.

What it restates (reference paths relative to /root/reference):

* the gate set the EVQE genome emits -- ``id``, ``u(theta, phi, lam)`` on one qubit and
  ``CU3Gate(theta, phi, lam)`` on ``(control, target)``
  (queasars/minimum_eigensolvers/evqe/quantum_circuit/quantum_gate.py:78-79, :96-102, :157-165);
  the matrices themselves are Qiskit's published definitions (UGate / CU3Gate docs);
* little-endian qubit order: qubit q is bit q of the basis-state index, a Pauli label's rightmost
  character acts on qubit 0 (queasars/utility/pauli_strings.py:38-40,
  queasars/job_shop_scheduling/domain_wall_hamiltonian_encoder.py:121);
* what ``OperatorCircuitEvaluator.evaluate_circuits`` returns: real(<psi|H|psi>) for the state
  prepared from |0...0> (queasars/circuit_evaluation/circuit_evaluation.py:200-215);
* what ``OperatorSamplerCircuitEvaluator`` computes from a measured distribution: the expectation or
  CVaR_alpha of a *diagonal* SparsePauliOp
  (queasars/circuit_evaluation/expectation_calculation.py:14-32, :35-69), where the value of a basis
  state is sum_k coeff_k * (-1)^popcount(state & z_k) (qiskit_algorithms ``_evaluate_sparsepauli``).

Two independent formulations are kept on purpose: a tensor-reshape one
(``simulate``/``pauli_expectation``) and a deliberately dumb dense-matrix one
(``simulate_dense``/``pauli_expectation_dense``, n <= 12).  They must agree to 1e-13.

An "op" here is a plain tuple ``(kind, target, control, theta, phi, lam)`` with bound angles:
kind 0 = id, 1 = u, 2 = cu3; control is -1 unless kind == 2.
"""

from __future__ import annotations

import math
from typing import Iterable, Sequence

import numpy as np

ID, U, CU3 = 0, 1, 2

_PAULI = {
    "I": np.array([[1, 0], [0, 1]], dtype=np.complex128),
    "X": np.array([[0, 1], [1, 0]], dtype=np.complex128),
    "Y": np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
    "Z": np.array([[1, 0], [0, -1]], dtype=np.complex128),
}


def u_matrix(theta: float, phi: float, lam: float) -> np.ndarray:
    """Qiskit's UGate: [[cos(t/2), -e^{i lam} sin(t/2)], [e^{i phi} sin(t/2), e^{i(phi+lam)} cos(t/2)]]."""
    c, s = math.cos(theta / 2.0), math.sin(theta / 2.0)
    return np.array(
        [
            [c, -np.exp(1j * lam) * s],
            [np.exp(1j * phi) * s, np.exp(1j * (phi + lam)) * c],
        ],
        dtype=np.complex128,
    )


# --------------------------------------------------------------------------------------------------
# Formulation A: tensor reshape
# --------------------------------------------------------------------------------------------------


def zero_state(n_qubits: int) -> np.ndarray:
    state = np.zeros(1 << n_qubits, dtype=np.complex128)
    state[0] = 1.0
    return state


def _apply_1q(state: np.ndarray, n: int, q: int, m: np.ndarray) -> np.ndarray:
    # index i = sum_q bit_q 2^q  =>  in C order qubit q is axis n-1-q
    psi = state.reshape((1 << (n - 1 - q), 2, 1 << q))
    out = np.empty_like(psi)
    out[:, 0, :] = m[0, 0] * psi[:, 0, :] + m[0, 1] * psi[:, 1, :]
    out[:, 1, :] = m[1, 0] * psi[:, 0, :] + m[1, 1] * psi[:, 1, :]
    return out.reshape(-1)


def _apply_controlled_1q(state: np.ndarray, n: int, control: int, target: int, m: np.ndarray) -> np.ndarray:
    idx = np.arange(1 << n, dtype=np.int64)
    sel0 = idx[((idx >> control) & 1 == 1) & ((idx >> target) & 1 == 0)]
    sel1 = sel0 | (1 << target)
    out = state.copy()
    a0, a1 = state[sel0], state[sel1]
    out[sel0] = m[0, 0] * a0 + m[0, 1] * a1
    out[sel1] = m[1, 0] * a0 + m[1, 1] * a1
    return out


def apply_op(state: np.ndarray, n: int, op: Sequence) -> np.ndarray:
    kind, target, control, theta, phi, lam = op
    if kind == ID:
        return state
    m = u_matrix(theta, phi, lam)
    if kind == U:
        return _apply_1q(state, n, int(target), m)
    if kind == CU3:
        if control == target or control < 0:
            raise ValueError("cu3 needs a control distinct from its target")
        return _apply_controlled_1q(state, n, int(control), int(target), m)
    raise ValueError(f"unknown op kind {kind}")


def simulate(n_qubits: int, ops: Iterable[Sequence], initial_state: np.ndarray | None = None) -> np.ndarray:
    """State after applying ``ops`` in order to |0...0> (or ``initial_state``)."""
    state = zero_state(n_qubits) if initial_state is None else np.asarray(initial_state, dtype=np.complex128).copy()
    for op in ops:
        state = apply_op(state, n_qubits, op)
    return state


def label_to_masks(label: str) -> tuple[int, int]:
    """Pauli label (rightmost char = qubit 0) -> (x_mask, z_mask); Y sets both."""
    x = z = 0
    n = len(label)
    for pos, ch in enumerate(label):
        q = n - 1 - pos
        if ch == "X":
            x |= 1 << q
        elif ch == "Z":
            z |= 1 << q
        elif ch == "Y":
            x |= 1 << q
            z |= 1 << q
        elif ch != "I":
            raise ValueError(f"bad Pauli character {ch!r}")
    return x, z


def _popcount_parity(v: np.ndarray) -> np.ndarray:
    v = v.copy()
    for s in (32, 16, 8, 4, 2, 1):
        v ^= v >> s
    return v & 1


def pauli_term_expectation(state: np.ndarray, x_mask: int, z_mask: int) -> complex:
    """<psi| P |psi> for P = i^{|x&z|} X^x Z^z  (so X, Y, Z tensor factors as written)."""
    n_amp = state.shape[0]
    idx = np.arange(n_amp, dtype=np.int64)
    src = idx ^ x_mask
    sign = 1.0 - 2.0 * _popcount_parity(src & z_mask)
    n_y = bin(x_mask & z_mask).count("1")
    phase = (1j) ** (n_y % 4)
    return complex(phase * np.sum(np.conj(state) * state[src] * sign))


def pauli_expectation(
    state: np.ndarray, x_masks: Sequence[int], z_masks: Sequence[int], coeffs: Sequence[complex]
) -> complex:
    total = 0.0 + 0.0j
    for x, z, c in zip(x_masks, z_masks, coeffs):
        total += complex(c) * pauli_term_expectation(state, int(x), int(z))
    return total


def diagonal_values(n_qubits: int, z_masks: Sequence[int], coeffs: Sequence[float]) -> np.ndarray:
    """D[i] = sum_k coeff_k (-1)^popcount(i & z_k): the value `_evaluate_sparsepauli` gives basis state i."""
    idx = np.arange(1 << n_qubits, dtype=np.int64)
    out = np.zeros(1 << n_qubits, dtype=np.float64)
    for z, c in zip(z_masks, coeffs):
        out += float(np.real(c)) * (1.0 - 2.0 * _popcount_parity(idx & int(z)))
    return out


def evaluate_sparsepauli(state_index: int, z_masks: Sequence[int], coeffs: Sequence[complex]) -> complex:
    """Value of one measured basis state under a diagonal operator (expectation_calculation.py:65)."""
    total = 0.0 + 0.0j
    for z, c in zip(z_masks, coeffs):
        total += complex(c) * (-1.0) ** bin(state_index & int(z)).count("1")
    return total


def probabilities(state: np.ndarray) -> np.ndarray:
    return (state.real**2 + state.imag**2).astype(np.float64)


# --------------------------------------------------------------------------------------------------
# Formulation B: explicit dense 2^n x 2^n operators.  Kept dumb on purpose.
# --------------------------------------------------------------------------------------------------


def _kron_all(factors_msb_first: Sequence[np.ndarray]) -> np.ndarray:
    out = np.array([[1.0 + 0.0j]])
    for f in factors_msb_first:
        out = np.kron(out, f)
    return out


def dense_op_matrix(n: int, op: Sequence) -> np.ndarray:
    kind, target, control, theta, phi, lam = op
    dim = 1 << n
    if kind == ID:
        return np.eye(dim, dtype=np.complex128)
    m = u_matrix(theta, phi, lam)
    eye = _PAULI["I"]
    if kind == U:
        return _kron_all([m if q == target else eye for q in range(n - 1, -1, -1)])
    p0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)
    p1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)
    a = _kron_all([p0 if q == control else eye for q in range(n - 1, -1, -1)])
    b = _kron_all([p1 if q == control else (m if q == target else eye) for q in range(n - 1, -1, -1)])
    return a + b


def simulate_dense(n_qubits: int, ops: Iterable[Sequence]) -> np.ndarray:
    if n_qubits > 12:
        raise ValueError("dense formulation is for n <= 12")
    state = zero_state(n_qubits)
    for op in ops:
        state = dense_op_matrix(n_qubits, op) @ state
    return state


def dense_pauli(label: str) -> np.ndarray:
    return _kron_all([_PAULI[ch] for ch in label])


def pauli_expectation_dense(state: np.ndarray, labels: Sequence[str], coeffs: Sequence[complex]) -> complex:
    total = 0.0 + 0.0j
    for label, c in zip(labels, coeffs):
        total += complex(c) * complex(np.vdot(state, dense_pauli(label) @ state))
    return total


# --------------------------------------------------------------------------------------------------
# Sampler branch post-processing
# --------------------------------------------------------------------------------------------------


def cvar_expectation(state_list: Sequence[tuple[object, float, float]], alpha: float) -> float:
    """CVaR accumulation, following expectation_calculation.py:14-32 step by step.

    ``state_list`` holds (state, probability, value).  For alpha != 1 the list is sorted by value, then
    probability mass is gathered until alpha is reached and the sum is divided by alpha.
    """
    items = list(state_list)
    if not np.isclose(alpha, 1):
        items = sorted(items, key=lambda t: t[2])
    gathered = 0.0
    expectation = 0.0
    for _, probability, value in items:
        probability = min(alpha - gathered, probability)
        expectation += probability * value
        gathered += probability
        if np.isclose(gathered, alpha):
            break
    return expectation / alpha


def expectation_from_distribution(
    distribution: dict[int, float], z_masks: Sequence[int], coeffs: Sequence[complex], alpha: float = 1.0
) -> float:
    """`get_expectation_with_operator` (expectation_calculation.py:35-69) for a {state: probability} dict."""
    if alpha <= 0 or 1 < alpha:
        raise ValueError("alpha must be in the range (0, 1]!")
    evaluations = [
        (state, p, evaluate_sparsepauli(int(state), z_masks, coeffs).real) for state, p in distribution.items()
    ]
    if np.isclose(alpha, 1):
        # qiskit.result.sampled_expectation_value: sum_i p_i * value_i
        return float(sum(p * v for _, p, v in evaluations))
    return float(cvar_expectation(evaluations, alpha))


def sample_counts(probs: np.ndarray, shots: int, seed: int) -> dict[int, int]:
    """Draw ``shots`` basis states from ``probs`` (oracle-side sampler; only statistically comparable)."""
    rng = np.random.default_rng(seed)
    cdf = np.cumsum(probs)
    cdf /= cdf[-1]
    draws = np.searchsorted(cdf, rng.random(shots), side="right")
    states, counts = np.unique(draws, return_counts=True)
    return {int(s): int(c) for s, c in zip(states, counts)}
