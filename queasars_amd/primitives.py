"""Primitive-shaped front ends: the two call shapes the reference's evaluators use on Qiskit primitives, served by the
GPU backend (SURVEY.md 8(b) "protocol B").

``queasars.circuit_evaluation`` only ever does

    estimator.run(pubs=((circuit, operator, values), ...), precision=p).result()   -> iterable of r with r.data.evs
    sampler.run(pubs=((circuit, values), ...), shots=s).result()                   -> iterable of r with
                                                                                      r.data["meas"].get_counts()

(circuit_evaluation.py:204-215 and :50-59).  The classes below answer exactly that, duck-typed: they do not subclass
Qiskit's ``BaseEstimatorV2`` / ``BaseSamplerV2`` (Qiskit is not a dependency), which the reference's evaluators do not
check.  Circuits may be :class:`~queasars_amd.ir.CircuitIR` or Qiskit ``QuantumCircuit`` objects, operators
:class:`~queasars_amd.ir.PauliOperator` or ``SparsePauliOp`` (converted by ``queasars_amd.qiskit_adapter`` and cached
per object).
"""

from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Iterable, Optional, Sequence

import numpy as np

from queasars_amd import qiskit_adapter
from queasars_amd.circuit_evaluation.circuit_evaluation import OperatorCircuitEvaluator, StatevectorDevice
from queasars_amd.ir import CircuitIR, PauliOperator


class _Job:
    def __init__(self, results: list):
        self._results = results

    def result(self) -> list:
        return self._results


class _Converter:
    """Caches the plain-data form of foreign circuit / operator objects by identity (the objects are kept alive)."""

    def __init__(self) -> None:
        self._circuits: dict[int, tuple[Any, CircuitIR]] = {}
        self._operators: dict[int, tuple[Any, PauliOperator]] = {}

    def circuit(self, c: Any) -> CircuitIR:
        if isinstance(c, CircuitIR):
            return c
        hit = self._circuits.get(id(c))
        if hit is None:
            hit = (c, qiskit_adapter.circuit_from_qiskit(c))
            self._circuits[id(c)] = hit
        return hit[1]

    def operator(self, op: Any) -> PauliOperator:
        if isinstance(op, PauliOperator):
            return op
        hit = self._operators.get(id(op))
        if hit is None:
            hit = (op, qiskit_adapter.operator_from_qiskit(op))
            self._operators[id(op)] = hit
        return hit[1]


class GpuEstimator:
    """``run(pubs, precision=...)`` over exact statevector expectation values; ``precision`` other than 0 / None adds
    Gaussian noise of that standard deviation (what an estimator's target precision means to the reference)."""

    def __init__(self, dtype: str = "fp64", device: int = 0, seed: Optional[int] = None):
        self._dtype, self._device_index = dtype, device
        self._rng = np.random.default_rng(seed)
        self._convert = _Converter()
        self._evaluators: dict[int, tuple[PauliOperator, OperatorCircuitEvaluator]] = {}

    def _evaluator(self, operator: PauliOperator) -> OperatorCircuitEvaluator:
        hit = self._evaluators.get(id(operator))
        if hit is None:
            hit = (operator, OperatorCircuitEvaluator(operator, dtype=self._dtype, device=self._device_index))
            self._evaluators[id(operator)] = hit
        return hit[1]

    def run(self, pubs: Iterable[Sequence[Any]], *, precision: Optional[float] = None) -> _Job:
        pubs = [tuple(pub) for pub in pubs]
        out: list[Optional[float]] = [None] * len(pubs)
        by_operator: dict[int, list[int]] = {}
        operators = []
        for i, pub in enumerate(pubs):
            operators.append(self._convert.operator(pub[1]))
            by_operator.setdefault(id(operators[-1]), []).append(i)
        for indices in by_operator.values():  # one batched call per distinct operator
            evaluator = self._evaluator(operators[indices[0]])
            circuits = [self._convert.circuit(pubs[i][0]) for i in indices]
            values = [list(np.ravel(pubs[i][2])) if len(pubs[i]) > 2 and pubs[i][2] is not None else [] for i in indices]
            for i, value in zip(indices, evaluator.evaluate_circuits(circuits, values)):
                out[i] = value
        if precision:
            out = [v + float(self._rng.normal(0.0, precision)) for v in out]
        return _Job([SimpleNamespace(data=SimpleNamespace(evs=np.asarray(v)), metadata={"target_precision": precision or 0.0})
                     for v in out])


class _BitArray:
    def __init__(self, states: np.ndarray, n_qubits: int):
        self._states, self._n = states, n_qubits

    def get_counts(self) -> dict[str, int]:
        values, counts = np.unique(self._states, return_counts=True)
        return {format(int(v), f"0{self._n}b"): int(c) for v, c in zip(values, counts)}

    def get_int_counts(self) -> dict[int, int]:
        values, counts = np.unique(self._states, return_counts=True)
        return {int(v): int(c) for v, c in zip(values, counts)}


class GpuSampler:
    """``run(pubs, shots=...)``: seeded inverse-CDF sampling of every qubit on the device; results expose
    ``data["meas"].get_counts()`` with Qiskit's bitstring convention (leftmost character = highest qubit)."""

    def __init__(self, n_qubits: int, dtype: str = "fp64", device: int = 0, seed: int = 0):
        self._device = StatevectorDevice(n_qubits, dtype=dtype, device=device)
        self._convert = _Converter()
        self._seed = int(seed)
        self._calls = 0

    def run(self, pubs: Iterable[Sequence[Any]], *, shots: int = 1024) -> _Job:
        pubs = [tuple(pub) for pub in pubs]
        circuits = [self._convert.circuit(pub[0]) for pub in pubs]
        values = [list(np.ravel(pub[1])) if len(pub) > 1 and pub[1] is not None else [] for pub in pubs]
        self._calls += 1
        states, _ = self._device.sample_batch(circuits, values, shots, seed=self._seed + self._calls)
        n = self._device.n_qubits
        return _Job([SimpleNamespace(data={"meas": _BitArray(np.asarray(row), n)}, metadata={"shots": shots}) for row in states])
