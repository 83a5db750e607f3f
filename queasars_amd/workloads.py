"""Synthetic workloads of SURVEY.md 8(d): EVQE populations from the restated genome, random Pauli / Ising operators.

Used by bench.py, the measurement scripts and the tests; no oracle, no device."""

from __future__ import annotations

import numpy as np

from queasars_amd.evqe import EVQEPopulation
from queasars_amd.ir import PauliOperator


def random_pauli_operator(n_qubits: int, n_terms: int, seed: int, alphabet: str = "IXYZ") -> PauliOperator:
    """T distinct random Pauli strings (no all-identity string), coefficients uniform(-1, 1)."""
    rng = np.random.default_rng(seed)
    labels: list[str] = []
    seen = set()
    while len(labels) < n_terms:
        label = "".join(rng.choice(list(alphabet), size=n_qubits))
        if set(label) == {"I"} or label in seen:
            if 4**n_qubits - 1 <= len(seen):
                break
            continue
        seen.add(label)
        labels.append(label)
    coeffs = rng.uniform(-1.0, 1.0, size=len(labels))
    return PauliOperator(labels, coeffs)


def random_ising_operator(n_qubits: int, seed: int) -> PauliOperator:
    """H = sum_{i<j} J_ij Z_i Z_j + sum_i h_i Z_i, J, h ~ N(0, 1)."""
    rng = np.random.default_rng(seed)
    terms = []
    for i in range(n_qubits):
        for j in range(i + 1, n_qubits):
            terms.append(("ZZ", [i, j], float(rng.normal())))
    for i in range(n_qubits):
        terms.append(("Z", [i], float(rng.normal())))
    return PauliOperator.from_sparse_list(terms, n_qubits)


def population_circuits(n_qubits: int, n_layers: int, n_individuals: int, seed: int, randomize: bool = True):
    pop = EVQEPopulation.random_population(n_qubits, n_layers, n_individuals, randomize, seed)
    circuits = [ind.get_parameterized_quantum_circuit() for ind in pop.individuals]
    params = [list(ind.parameter_values) for ind in pop.individuals]
    return pop, circuits, params
