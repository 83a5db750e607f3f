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
    """Expectation (alpha == 1) or CVaR_alpha of ``(state, probability, value)`` entries: the mean of ``value`` over
    the probability mass ``alpha`` with the lowest values (reference [14-32]).  Vectorised: sort by value, take the
    cumulative mass, clip it at ``alpha``; the increments of the clipped curve are the weights."""
    probabilities = np.fromiter((entry[1] for entry in state_list), dtype=np.float64, count=len(state_list))
    values = np.fromiter((entry[2] for entry in state_list), dtype=np.float64, count=len(state_list))
    if not np.isclose(alpha, 1):
        order = np.argsort(values, kind="stable")
        probabilities, values = probabilities[order], values[order]
    clipped = np.minimum(np.cumsum(probabilities), alpha)
    weights = np.diff(clipped, prepend=0.0)
    return float(np.dot(weights, values) / alpha)


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
