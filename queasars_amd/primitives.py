"""Primitive-shaped front ends: the two call shapes the reference's evaluators use on Qiskit primitives, served by the
GPU backend (SURVEY.md 8(b) "protocol B").

``queasars.circuit_evaluation`` only ever does

    estimator.run(pubs=((circuit, operator, values), ...), precision=p).result()   -> iterable of r with r.data.evs
    sampler.run(pubs=((circuit, values), ...), shots=s).result()                   -> iterable of r with
                                                                                      r.data["meas"].get_counts()

(circuit_evaluation.py:204-215 and :50-59).  The classes below answer exactly that.  Where Qiskit is importable they
subclass ``BaseEstimatorV2`` / ``BaseSamplerV2``; where it is not (this repository's environment) they are plain classes
of the same shape, which is all the reference's evaluators rely on.  Circuits may be :class:`~queasars_amd.ir.CircuitIR` or Qiskit ``QuantumCircuit`` objects, operators
:class:`~queasars_amd.ir.PauliOperator` or ``SparsePauliOp`` (converted by ``queasars_amd.qiskit_adapter`` and cached
per object).
"""

from __future__ import annotations

import weakref
from collections import OrderedDict
from types import SimpleNamespace
from typing import Any, Iterable, Optional, Sequence

import numpy as np

from queasars_amd import qiskit_adapter
from queasars_amd.circuit_evaluation.circuit_evaluation import OperatorCircuitEvaluator, StatevectorDevice
from queasars_amd.ir import CircuitIR, PauliOperator


# On a host that has Qiskit the two front ends ARE Qiskit V2 primitives (subclasses of the abstract bases, so that
# isinstance checks and type annotations of user code hold); without Qiskit they are plain classes of the same shape.
try:  # pragma: no cover - Qiskit is not installable in this repository's environment
    from qiskit.primitives import BaseEstimatorV2 as _EstimatorBase
    from qiskit.primitives import BaseSamplerV2 as _SamplerBase
except Exception:  # ModuleNotFoundError here
    _EstimatorBase = _SamplerBase = object


class _Job:
    def __init__(self, results: list):
        self._results = results

    def result(self) -> list:
        return self._results


class _IdentityCache:
    """Conversion results keyed by object identity WITHOUT pinning the objects: an entry holds its key object weakly
    and disappears with it (so does the converted circuit, and with it the device-side plan); objects that cannot be
    weakly referenced are held in a small bounded table instead.  The reference creates fresh circuits on almost every
    call (``measure_all(inplace=False)``, one circuit per individual and generation), so an unbounded identity cache
    would only ever grow."""

    def __init__(self, convert, limit: int = 512):
        self._convert, self._limit = convert, int(limit)
        self._weak: dict[int, tuple[weakref.ref, Any]] = {}
        self._strong: dict[int, tuple[Any, Any]] = {}

    def __len__(self) -> int:
        return len(self._weak) + len(self._strong)

    def get(self, obj: Any) -> Any:
        key = id(obj)
        hit = self._weak.get(key)
        if hit is not None and hit[0]() is obj:
            return hit[1]
        hit = self._strong.get(key)
        if hit is not None and hit[0] is obj:
            return hit[1]
        value = self._convert(obj)
        try:
            table = self._weak
            table[key] = (weakref.ref(obj, lambda _r, k=key, t=table: t.pop(k, None)), value)
        except TypeError:
            if len(self._strong) >= self._limit:
                for old in list(self._strong)[: self._limit // 2]:
                    self._strong.pop(old, None)
            self._strong[key] = (obj, value)
        return value


class _Converter:
    """Plain-data form of foreign circuit / operator objects (see :class:`_IdentityCache`)."""

    def __init__(self) -> None:
        self._circuits = _IdentityCache(qiskit_adapter.circuit_from_qiskit)
        self._operators = _IdentityCache(qiskit_adapter.operator_from_qiskit)

    def circuit(self, c: Any) -> CircuitIR:
        return c if isinstance(c, CircuitIR) else self._circuits.get(c)

    def operator(self, op: Any) -> PauliOperator:
        return op if isinstance(op, PauliOperator) else self._operators.get(op)


def _operator_key(op: PauliOperator) -> tuple:
    return (op.num_qubits, op.x_mask.tobytes(), op.z_mask.tobytes(), op.coeffs.tobytes())


class GpuEstimator(_EstimatorBase):
    """``run(pubs, precision=...)`` over exact statevector expectation values; ``precision`` other than 0 / None adds
    Gaussian noise of that standard deviation (what an estimator's target precision means to the reference).

    One :class:`StatevectorDevice` per qubit count serves every operator (its tables are rebuilt when the operator
    changes); operators are recognised by content, not identity, and at most ``max_operators`` evaluators are kept."""

    def __init__(self, dtype: str = "fp64", device: int = 0, seed: Optional[int] = None, max_operators: int = 8):
        self._dtype, self._device_index, self._seed = dtype, device, seed
        self._rng = np.random.default_rng(seed)
        self._convert = _Converter()
        self._devices: dict[int, StatevectorDevice] = {}
        self._evaluators: "OrderedDict[tuple, OperatorCircuitEvaluator]" = OrderedDict()
        self._max_operators = max(1, int(max_operators))

    def backend_options(self) -> dict:
        """dtype / device / seed, for ``configured_primitives.evaluator_for``."""
        return {"dtype": self._dtype, "device": self._device_index, "seed": self._seed}

    def _evaluator(self, operator: PauliOperator) -> OperatorCircuitEvaluator:
        key = _operator_key(operator)
        hit = self._evaluators.get(key)
        if hit is None:
            dev = self._devices.get(operator.num_qubits)
            if dev is None:
                dev = StatevectorDevice(operator.num_qubits, dtype=self._dtype, device=self._device_index)
                self._devices[operator.num_qubits] = dev
            hit = OperatorCircuitEvaluator(operator, statevector_device=dev)
            self._evaluators[key] = hit
            while len(self._evaluators) > self._max_operators:
                self._evaluators.popitem(last=False)
        else:
            self._evaluators.move_to_end(key)
        return hit

    def run(self, pubs: Iterable[Sequence[Any]], *, precision: Optional[float] = None) -> _Job:
        pubs = [tuple(pub) for pub in pubs]
        out: list[Optional[float]] = [None] * len(pubs)
        by_operator: dict[tuple, list[int]] = {}
        operators = []
        for i, pub in enumerate(pubs):
            operators.append(self._convert.operator(pub[1]))
            by_operator.setdefault(_operator_key(operators[-1]), []).append(i)
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


class GpuSampler(_SamplerBase):
    """``run(pubs, shots=...)``: seeded inverse-CDF sampling of every qubit on the device; results expose
    ``data["meas"].get_counts()`` with Qiskit's bitstring convention (leftmost character = highest qubit)."""

    def __init__(self, n_qubits: int, dtype: str = "fp64", device: int = 0, seed: int = 0):
        self._device = StatevectorDevice(n_qubits, dtype=dtype, device=device)
        self._options = {"dtype": dtype, "device": device, "seed": int(seed)}
        self._convert = _Converter()
        self._seed = int(seed)
        self._calls = 0

    def backend_options(self) -> dict:
        """dtype / device / seed, for ``configured_primitives.evaluator_for``."""
        return dict(self._options)

    def run(self, pubs: Iterable[Sequence[Any]], *, shots: Optional[int] = None) -> _Job:
        shots = 1024 if shots is None else int(shots)
        pubs = [tuple(pub) for pub in pubs]
        circuits = [self._convert.circuit(pub[0]) for pub in pubs]
        values = [list(np.ravel(pub[1])) if len(pub) > 1 and pub[1] is not None else [] for pub in pubs]
        self._calls += 1
        states, _ = self._device.sample_batch(circuits, values, shots, seed=self._seed + self._calls)
        n = self._device.n_qubits
        return _Job([SimpleNamespace(data={"meas": _BitArray(np.asarray(row), n)}, metadata={"shots": shots}) for row in states])
