"""Expectation / CVaR of a measured distribution (host side of the sampler branch).

Restates queasars/circuit_evaluation/expectation_calculation.py: the CVaR accumulation [14-32],
``get_expectation_with_operator`` [35-69] and ``get_expectation_with_bitstring_evaluator`` [72-103].
The value of a measured basis state under a diagonal operator is
``sum_k coeff_k * (-1)^popcount(state & z_k)`` (what qiskit_algorithms' ``_evaluate_sparsepauli`` returns,
reference [65]); for alpha == 1 the reference calls ``qiskit.result.sampled_expectation_value`` [60-61],
which is the probability-weighted mean of the same values.

These run over at most ``shots`` distinct outcomes, so they stay on the host.
"""

from __future__ import annotations

from typing import Any, Mapping, Optional

import numpy as np

from queasars_amd.circuit_evaluation.bitstring_evaluation import BitstringEvaluator
from queasars_amd.ir import PauliOperator


def _check_alpha(alpha: float) -> None:
    if alpha <= 0 or 1 < alpha:
        raise ValueError("alpha must be in the range (0, 1]!")


def _get_expectation(state_list: list[tuple[Any, float, float]], alpha: float) -> float:
    """Gather probability mass in order (ascending value when alpha != 1) until alpha is reached."""
    if not np.isclose(alpha, 1):
        state_list = sorted(state_list, key=lambda entry: entry[2])
    gathered = 0.0
    expectation = 0.0
    for _, probability, value in state_list:
        probability = min(alpha - gathered, probability)
        expectation += probability * value
        gathered += probability
        if np.isclose(gathered, alpha):
            break
    return expectation / alpha


def basis_state_values(states: np.ndarray, operator: PauliOperator) -> np.ndarray:
    """real(sum_k coeff_k (-1)^popcount(state & z_k)) for every entry of ``states`` (vectorised)."""
    states = np.asarray(states, dtype=np.uint64)
    values = np.zeros(states.shape, dtype=np.float64)
    for z, c in zip(operator.z_mask, operator.coeffs):
        v = states & z
        for shift in (32, 16, 8, 4, 2, 1):
            v = v ^ (v >> np.uint64(shift))
        values += c.real * (1.0 - 2.0 * (v & np.uint64(1)).astype(np.float64))
    return values


def get_expectation_with_operator(
    measurement_distribution: Mapping[int, float], operator: PauliOperator, alpha: float = 1
) -> float:
    _check_alpha(alpha)
    if not operator.is_diagonal():
        raise ValueError("The operator must be diagonal (only I and Z factors)!")
    states = np.fromiter(measurement_distribution.keys(), dtype=np.uint64, count=len(measurement_distribution))
    probabilities = np.fromiter(measurement_distribution.values(), dtype=np.float64, count=len(measurement_distribution))
    values = basis_state_values(states, operator)
    if np.isclose(alpha, 1):
        return float(np.sum(probabilities * values))
    evaluations = [(int(s), float(p), float(v)) for s, p, v in zip(states, probabilities, values)]
    return float(_get_expectation(evaluations, alpha))


def get_expectation_with_bitstring_evaluator(
    measurement_distribution: Mapping[int, float],
    bitstring_evaluator: BitstringEvaluator,
    alpha: float = 1,
    n_qubits: Optional[int] = None,
) -> float:
    _check_alpha(alpha)
    width = bitstring_evaluator.input_length if n_qubits is None else n_qubits
    evaluations = []
    for state, probability in measurement_distribution.items():
        bitstring = format(int(state), f"0{width}b")  # qiskit's binary_probabilities(): qubit 0 is the last char
        evaluations.append((bitstring, probability, bitstring_evaluator.evaluate_bitstring(bitstring)))
    return float(_get_expectation(evaluations, alpha))
