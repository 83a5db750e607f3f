"""Plain-data circuit and operator types that cross the C ABI.

The reference hands ``qiskit.QuantumCircuit`` / ``SparsePauliOp`` objects to a Qiskit primitive
(reference: queasars/circuit_evaluation/circuit_evaluation.py:200-215).  Qiskit is not a dependency of
this backend, so the hot path works on two plain records instead:

* :class:`CircuitIR` -- the decomposed EVQE circuit: a list of ``id`` / ``u`` / ``cu3`` ops
  (the only instructions left after ``decompose()``, reference:
  queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/individual.py:288-322), each angle being
  either a bound literal or an explicit index into the call's flat parameter list.  The index is data, not
  convention: Qiskit binds a flat list to ``circuit.parameters``, which is sorted by parameter *name*
  ("Parameter-binding order"), and whoever builds the IR resolves that order.
* :class:`PauliOperator` -- a sparse Pauli sum as (x_mask, z_mask, coeff) triples with Qiskit's label
  convention (rightmost character = qubit 0, reference: queasars/utility/pauli_strings.py:38-40).

Both are laid out exactly as ``include/qsv.h`` declares them, so they are passed to ``libqsv`` without
conversion.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Iterable, Optional, Sequence, Union

import numpy as np

OP_ID, OP_U, OP_CU3 = 0, 1, 2
NO_CONTROL = 0xFF

# struct qsv_op in include/qsv.h (40 bytes, natural alignment)
QSV_OP_DTYPE = np.dtype(
    [
        ("kind", np.uint8),
        ("target", np.uint8),
        ("control", np.uint8),
        ("flags", np.uint8),
        ("p_theta", np.int32),
        ("p_phi", np.int32),
        ("p_lambda", np.int32),
        ("theta", np.float64),
        ("phi", np.float64),
        ("lam", np.float64),
    ],
    align=True,
)
assert QSV_OP_DTYPE.itemsize == 40
_ROW = struct.Struct("<BBBBiiiddd")  # the same record, for appending one op at a time
assert _ROW.size == 40


@dataclass(frozen=True)
class ParamRef:
    """Reference to entry ``index`` of the flat parameter list an evaluation is called with."""

    index: int


Angle = Union[float, ParamRef]


class CircuitIR:
    """A decomposed EVQE circuit on ``n_qubits`` qubits.

    Build it with :meth:`id`, :meth:`u` and :meth:`cu3` (argument order as Qiskit's
    ``QuantumCircuit.u(theta, phi, lam, qubit)`` and ``CU3Gate(theta, phi, lam)`` on ``(control, target)``).
    """

    edits_of_registered = 0  # counts edits of circuits some device had registered (see _append)

    def __init__(self, n_qubits: int):
        if not 1 <= int(n_qubits) <= 34:
            raise ValueError("n_qubits must be in [1, 34]")
        self._n_qubits = int(n_qubits)
        self._rows: list[tuple] = []
        self._bytes = bytearray()  # the rows as qsv_op records (what packed() hands to the library)
        self._n_parameters = 0
        self._packed: Optional[np.ndarray] = None
        self._version = 0  # bumped by every edit (caches of derived circuits compare it)
        # (device key -> circuit id) cache used by the evaluator; cleared on mutation.  Process-local: a device key
        # is only meaningful in the process that created the device, so copies and pickles start without it.
        self._registered: dict[object, int] = {}
        # A state kept on a device (StatevectorDevice.keep_states) this circuit starts from instead of |0..0>: set by
        # continue_from.  Such a circuit belongs to that device and to this process.
        self._kept_state = None

    # -- copying ----------------------------------------------------------------------------------
    def __getstate__(self) -> dict:
        if self._kept_state is not None:
            raise TypeError("a circuit that continues a state kept on a device cannot be pickled (the state does not travel)")
        state = self.__dict__.copy()
        state["_registered"] = {}
        state["_packed"] = None
        return state

    def __deepcopy__(self, memo) -> "CircuitIR":
        out = CircuitIR(self._n_qubits)
        out._rows = list(self._rows)  # rows are immutable tuples
        out._bytes = bytearray(self._bytes)
        out._n_parameters = self._n_parameters
        out._kept_state = self._kept_state
        memo[id(self)] = out
        return out

    __copy__ = lambda self: self.__deepcopy__({})  # noqa: E731

    # -- construction ---------------------------------------------------------------------------
    def _angle(self, a: Angle) -> tuple[int, float]:
        if isinstance(a, ParamRef):
            if a.index < 0:
                raise ValueError("parameter index must be >= 0")
            self._n_parameters = max(self._n_parameters, a.index + 1)
            return a.index, 0.0
        return -1, float(a)

    def _check_qubit(self, q: int) -> int:
        q = int(q)
        if not 0 <= q < self._n_qubits:
            raise ValueError(f"qubit {q} out of range for {self._n_qubits} qubits")
        return q

    def _append(self, kind: int, target: int, control: int, theta: Angle, phi: Angle, lam: Angle) -> "CircuitIR":
        (it, vt), (ip, vp), (il, vl) = self._angle(theta), self._angle(phi), self._angle(lam)
        row = (kind, target, control, 0, it, ip, il, vt, vp, vl)
        self._rows.append(row)
        self._bytes += _ROW.pack(*row)
        self._packed = None
        self._version += 1
        if self._registered:
            # a circuit that a device already knows is being edited: devices must register it again, and any batch
            # metadata they cached by object identity is void
            CircuitIR.edits_of_registered += 1
            self._registered = {}
        return self

    @classmethod
    def from_rows(cls, n_qubits: int, rows: list, n_parameters: int) -> "CircuitIR":
        """A circuit from complete ``qsv_op`` rows ``(kind, target, control, 0, p_theta, p_phi, p_lambda, theta, phi, lam)``
        (control = NO_CONTROL for none; parameter indices -1 for literal angles) and its parameter count -- for a caller
        that produces valid rows wholesale (the EVQE genome); qubit indices are still checked."""
        out = cls(n_qubits)
        for row in rows:
            if not (0 <= row[1] < out._n_qubits and (row[2] == NO_CONTROL or (0 <= row[2] < out._n_qubits and row[2] != row[1]))):
                raise ValueError("qubit index out of range")
        out._rows = list(rows)
        out._bytes = bytearray(b"".join([_ROW.pack(*row) for row in rows]))
        out._n_parameters = int(n_parameters)
        return out

    def id(self, qubit: int) -> "CircuitIR":
        return self._append(OP_ID, self._check_qubit(qubit), NO_CONTROL, 0.0, 0.0, 0.0)

    def u(self, theta: Angle, phi: Angle, lam: Angle, qubit: int) -> "CircuitIR":
        return self._append(OP_U, self._check_qubit(qubit), NO_CONTROL, theta, phi, lam)

    def cu3(self, theta: Angle, phi: Angle, lam: Angle, control_qubit: int, target_qubit: int) -> "CircuitIR":
        c, t = self._check_qubit(control_qubit), self._check_qubit(target_qubit)
        if c == t:
            raise ValueError("control and target must differ")
        return self._append(OP_CU3, t, c, theta, phi, lam)

    def declare_parameters(self, n_parameters: int) -> "CircuitIR":
        """State that an evaluation is called with at least ``n_parameters`` values even if the highest-indexed ones are
        not used by any gate (a Qiskit circuit's ``parameters`` may hold such entries)."""
        if n_parameters > self._n_parameters:
            self._n_parameters = int(n_parameters)
            self._version += 1
            if self._registered:
                CircuitIR.edits_of_registered += 1
                self._registered = {}
            self._packed = None
        return self

    def compose(self, other: "CircuitIR") -> "CircuitIR":
        """New circuit: ``self`` followed by ``other`` (parameter indices of both are kept as they are)."""
        if other.n_qubits != self.n_qubits:
            raise ValueError("qubit counts differ")
        out = CircuitIR(self._n_qubits)
        out._rows = list(self._rows) + list(other._rows)
        out._bytes = self._bytes + other._bytes
        out._n_parameters = max(self._n_parameters, other._n_parameters)
        if other._kept_state is not None:
            raise ValueError("a circuit that continues a kept state cannot follow another circuit")
        out._kept_state = self._kept_state
        return out

    def continue_from(self, kept_state) -> "CircuitIR":
        """This circuit applied to ``kept_state`` (one of ``StatevectorDevice.keep_states(...)``) instead of to |0..0>: what a
        layer search evaluates again and again once everything in front of the searched layer is a state on the device
        (reference: optimize_layer_of_individual binds the other layers into the circuit, mutation.py:57-59).  Returns self."""
        if kept_state is not None and kept_state.n_qubits != self._n_qubits:
            raise ValueError("the kept state has another number of qubits")
        if self._registered:
            CircuitIR.edits_of_registered += 1
            self._registered = {}
        self._kept_state = kept_state
        self._version += 1
        return self

    @property
    def kept_state(self):
        return self._kept_state

    # -- views ----------------------------------------------------------------------------------
    @property
    def n_qubits(self) -> int:
        return self._n_qubits

    @property
    def num_qubits(self) -> int:  # qiskit spelling
        return self._n_qubits

    @property
    def num_parameters(self) -> int:
        return self._n_parameters

    def __len__(self) -> int:
        return len(self._rows)

    def packed(self) -> np.ndarray:
        """The ops as a contiguous ``qsv_op`` array (cached)."""
        if self._packed is None:
            self._packed = np.frombuffer(bytes(self._bytes), dtype=QSV_OP_DTYPE)
        return self._packed

    def bound_ops(self, parameter_values: Sequence[float]) -> list[tuple]:
        """``(kind, target, control, theta, phi, lam)`` tuples with every angle bound (control -1 if none)."""
        if len(parameter_values) < self._n_parameters:
            raise ValueError(f"circuit needs {self._n_parameters} parameter values, got {len(parameter_values)}")
        out = []
        for kind, target, control, _f, it, ip, il, vt, vp, vl in self._rows:
            theta = float(parameter_values[it]) if it >= 0 else vt
            phi = float(parameter_values[ip]) if ip >= 0 else vp
            lam = float(parameter_values[il]) if il >= 0 else vl
            out.append((kind, target, -1 if control == NO_CONTROL else control, theta, phi, lam))
        return out

    def depth(self) -> int:
        """Length of the longest chain of ops that follow each other on some qubit, every op (``id`` included) counting one --
        what ``QuantumCircuit.depth()`` returns for the same op list (the reference's tests hold an individual's circuit to
        one level per layer, test_evqe_individual.py:325-343)."""
        level = [0] * self._n_qubits
        for row in self._rows:
            target, control = row[1], row[2]
            if control == NO_CONTROL:
                level[target] += 1
            else:
                level[target] = level[control] = max(level[target], level[control]) + 1
        return max(level, default=0)

    def count_ops(self) -> dict[str, int]:
        names = {OP_ID: "id", OP_U: "u", OP_CU3: "cu3"}
        out: dict[str, int] = {}
        for row in self._rows:
            out[names[row[0]]] = out.get(names[row[0]], 0) + 1
        return out


class PauliOperator:
    """Sparse Pauli sum ``sum_k coeff_k * P_k`` (plain-data stand-in for qiskit's ``SparsePauliOp``)."""

    def __init__(self, labels: Sequence[str], coeffs: Optional[Iterable[complex]] = None):
        labels = list(labels)
        if not labels:
            raise ValueError("at least one Pauli label is required")
        n = len(labels[0])
        if n < 1 or n > 34:
            raise ValueError("labels must have between 1 and 34 characters")
        xs, zs = [], []
        for label in labels:
            if len(label) != n:
                raise ValueError("all labels must have the same length")
            x = z = 0
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
                    raise ValueError(f"invalid Pauli character {ch!r} in {label!r}")
            xs.append(x)
            zs.append(z)
        cs = np.ones(len(labels), dtype=np.complex128) if coeffs is None else np.asarray(list(coeffs), dtype=np.complex128)
        if cs.shape != (len(labels),):
            raise ValueError("need exactly one coefficient per label")
        self._n = n
        self._labels = labels
        self.x_mask = np.asarray(xs, dtype=np.uint64)
        self.z_mask = np.asarray(zs, dtype=np.uint64)
        self.coeffs = cs

    @classmethod
    def from_list(cls, terms: Sequence[tuple[str, complex]]) -> "PauliOperator":
        return cls([t[0] for t in terms], [t[1] for t in terms])

    @classmethod
    def from_sparse_list(cls, terms: Sequence[tuple[str, Sequence[int], complex]], num_qubits: int) -> "PauliOperator":
        """``[("ZZ", [0, 3], 0.5), ...]`` as in qiskit's ``SparsePauliOp.from_sparse_list``."""
        labels, coeffs = [], []
        for paulis, qubits, coeff in terms:
            chars = ["I"] * num_qubits
            for ch, q in zip(paulis, qubits):
                chars[num_qubits - 1 - q] = ch
            labels.append("".join(chars))
            coeffs.append(coeff)
        return cls(labels, coeffs)

    @property
    def num_qubits(self) -> int:
        return self._n

    @property
    def labels(self) -> list[str]:
        return list(self._labels)

    def __len__(self) -> int:
        return len(self._labels)

    def is_diagonal(self) -> bool:
        """Only I/Z factors (what the sampler branch requires, reference: expectation_calculation.py:35-69)."""
        return not bool(np.any(self.x_mask))

    def __reduce__(self):
        return (PauliOperator, (self._labels, self.coeffs.tolist()))
