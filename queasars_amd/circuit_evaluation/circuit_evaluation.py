"""GPU circuit evaluators behind the reference's evaluator protocol.

Mirrors ``queasars.circuit_evaluation.circuit_evaluation`` (reference file, lines in brackets):

* :class:`BaseCircuitEvaluator` -- ``evaluate_circuits(circuits, parameter_values) -> list[float]`` and
  ``n_qubits`` [62-87];
* :class:`CircuitEvaluatorException` [90];
* :class:`OperatorCircuitEvaluator` -- exact ``real(<psi|H|psi>)``; replaces the estimator branch [164-219],
  the Qiskit ``EstimatorV2`` and the batching mutex the reference needs around it
  (queasars/circuit_evaluation/mutex_primitives.py:25-199): the whole batch goes to the device in one call;
* :class:`OperatorSamplerCircuitEvaluator` -- expectation / CVaR of a diagonal operator from ``shots`` samples
  [94-161]; :class:`BitstringCircuitEvaluator` [222-291]; :func:`measure_quasi_distributions` [29-59].

Circuits are :class:`queasars_amd.ir.CircuitIR` objects (the decomposed ``id``/``u``/``cu3`` form the reference
evaluates), the operator is a :class:`queasars_amd.ir.PauliOperator`.  Results are ordered by input index
[68-70].  Constructor misuse raises ``ValueError`` with the reference's messages [128-131, 136-137, 140-144];
anything that goes wrong on the device raises :class:`CircuitEvaluatorException`.

There is no CPU fallback: if ``libqsv`` cannot be loaded or no GPU is present, construction fails.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
import uuid
import weakref
from itertools import count
from abc import ABC, abstractmethod
from typing import Optional, Sequence

from array import array

import numpy as np

from queasars_amd import _lib
from queasars_amd.circuit_evaluation.bitstring_evaluation import BitstringEvaluator
from queasars_amd.circuit_evaluation.expectation_calculation import (
    get_expectation_with_bitstring_evaluator,
    get_expectation_with_operator,
)
from queasars_amd.ir import QSV_OP_DTYPE, CircuitIR, PauliOperator


class CircuitEvaluatorException(Exception):
    """Class for exceptions caused during the evaluation of quantum circuits"""


class BaseCircuitEvaluator(ABC):
    """Abstract class to allow a seamless exchange of circuit evaluation methods in QUEASARS eigensolvers"""

    @abstractmethod
    def evaluate_circuits(self, circuits: list[CircuitIR], parameter_values: list[list[float]]) -> list[float]:
        """Circuit i is evaluated for parameter_values[i]; result i is at index i of the returned list."""

    @property
    @abstractmethod
    def n_qubits(self) -> int:
        """Size (in qubits) of the circuits this evaluator can evaluate."""


_device_serial = count(1)
_PROCESS_TOKEN = uuid.uuid4().hex  # registrations are process-local: a pickled circuit must not match here by accident


_pyhelp = None


def _load_pyhelp():
    """csrc/pyhelp.c through ctypes.PyDLL (the GIL stays held: it walks Python lists); None when it was not built."""
    global _pyhelp
    if _pyhelp is None:
        from queasars_amd import _build

        _pyhelp = False
        try:
            path = _build.build_pyhelp()  # (rebuilt when csrc/pyhelp.c or libqsv.so is newer; None when it cannot be)
            if path is None and _build.PYHELP_PATH.exists() and not _build.pyhelp_stale():
                path = _build.PYHELP_PATH
            if path is not None:
                lib = C.PyDLL(str(path))
                lib.qsv_pack_vectors.restype = C.c_ssize_t
                lib.qsv_pack_vectors.argtypes = [C.py_object, C.c_ssize_t, C.c_ssize_t, C.c_void_p, C.c_ssize_t]
                lib.qsv_py_expectation_values.restype = C.c_int
                lib.qsv_py_expectation_values.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.py_object,
                                                          C.c_void_p, C.c_ssize_t, C.c_void_p]
                lib.qsv_py_expectation_values_device.restype = C.c_int
                lib.qsv_py_expectation_values_device.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.py_object,
                                                                 C.c_void_p, C.c_ssize_t, C.c_void_p]
                lib.qsv_py_expectation_values_devparams.restype = C.c_int
                lib.qsv_py_expectation_values_devparams.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                                    C.c_void_p, C.c_void_p, C.c_void_p]
                lib.qsv_pack_exact.restype = C.c_ssize_t
                lib.qsv_pack_exact.argtypes = [C.py_object, C.c_ssize_t, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_ssize_t]
                _pyhelp = lib
        except OSError:
            _pyhelp = False
    return _pyhelp or None


_pyhelp_module = None


def _load_pyhelp_module():
    """csrc/pyhelp.c as an extension module (its ``eval_one``: a whole single-circuit call in C, no ctypes conversion);
    None when the helper was not built."""
    global _pyhelp_module
    if _pyhelp_module is None:
        _pyhelp_module = False
        if _load_pyhelp() is not None and not os.environ.get("QSV_LIBRARY"):
            try:
                import importlib.machinery
                import importlib.util

                from queasars_amd import _build

                loader = importlib.machinery.ExtensionFileLoader("_qsvpyhelp", str(_build.PYHELP_PATH))
                spec = importlib.util.spec_from_loader("_qsvpyhelp", loader)
                module = importlib.util.module_from_spec(spec)
                loader.exec_module(module)
                _pyhelp_module = module
            except (ImportError, OSError):
                _pyhelp_module = False
    return _pyhelp_module or None


def _has_none(seq) -> bool:
    """Some element IS None (by identity: a parameter vector may be a NumPy array, which answers ``==`` element by element)."""
    fast = _load_pyhelp_module()
    if fast is not None and isinstance(seq, (list, tuple)):
        return fast.has_none(seq)
    return any(v is None for v in seq)


def _pack_slice(vectors: Sequence[Sequence[float]], first: int, last: int, total: int) -> np.ndarray:
    """The parameter vectors ``vectors[first:last]`` back to back as one float64 array of ``total`` values."""
    helper = _load_pyhelp()
    if helper is None:
        return _pack_doubles(vectors[first:last], total)
    out = np.empty(max(total, 1), dtype=np.float64)
    n = helper.qsv_pack_vectors(vectors, first, last - first, out.ctypes.data, total)
    if n != total:
        raise ValueError("parameter vectors changed length while they were being packed")
    return out


def _pack_doubles(vectors: Sequence[Sequence[float]], total: int) -> np.ndarray:
    """Parameter vectors back to back as one float64 array.  ``array.fromlist`` is the fastest way CPython offers to
    turn lists of floats into doubles (about 15 ns per value, 40% less than ``np.fromiter`` over a chain)."""
    packed = array("d")
    for vec in vectors:
        if type(vec) is list:
            packed.fromlist(vec)
        elif isinstance(vec, np.ndarray):
            packed.frombytes(np.ascontiguousarray(vec, dtype=np.float64).tobytes())
        else:
            packed.extend(vec)
    if len(packed) != total:
        raise ValueError("parameter vectors changed length while they were being packed")
    return np.frombuffer(packed, dtype=np.float64)


def _make_gone(dead: list, watched: dict):
    """Callback of the weak references StatevectorDevice._watch creates: queue the dead circuit's id for destruction."""

    def gone(ref):
        entry = watched.pop(id(ref), None)
        if entry is not None:
            dead.append(entry[1])  # no library call here: see StatevectorDevice.__init__

    return gone


class KeptState:
    """A state kept resident on a device (``StatevectorDevice.keep_states``): what a layer search's evaluations have in common
    (reference: optimize_layer_of_individual, mutation.py:57-59).  Circuits made to start from it with
    ``CircuitIR.continue_from`` hold it; its memory on the device is reused once it and they are gone."""

    __slots__ = ("_owner", "_serial", "_id", "n_qubits", "__weakref__")

    def __init__(self, owner: "StatevectorDevice", prefix_id: int):
        self._owner = weakref.ref(owner)
        self._serial = owner._serial
        self._id = int(prefix_id)
        self.n_qubits = owner.n_qubits

    def release(self) -> None:
        """Let go now (idempotent).  Circuits that continue this state keep it alive on the device until they are gone."""
        pid, self._id = self._id, -1
        owner = self._owner()
        if pid >= 0 and owner is not None:
            owner._dead_states.append(pid)  # (destroyed at the next registration: never from inside an open batch)

    def __del__(self):
        try:
            self.release()
        except Exception:  # pragma: no cover
            pass


class StatevectorDevice:
    """Owns one ``qsv_t`` handle: the resident state buffers, plans and operator tables of one GPU.

    Thread safe (calls on a handle are serialised inside the library), picklable (re-created from plain data in
    the receiving process, as Dask-style executors need; reference: queasars/minimum_eigensolvers/evqe/evqe.py:39-44).
    """

    def __init__(
        self,
        n_qubits: int,
        dtype: str = "fp64",
        device: int = 0,
        tile_bits: int = 0,
        reg_bits: int = 0,
        low_bits: int = 0,
        group: int = 0,
        exchange: int = 0,
    ):
        if dtype not in ("fp64", "fp32"):
            raise ValueError("dtype must be 'fp64' or 'fp32'")
        self._args = (int(n_qubits), dtype, int(device), int(tile_bits), int(reg_bits), int(low_bits), int(group), int(exchange))
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        cfg = _lib.QsvPlanConfig(tile_bits, reg_bits, low_bits, group, exchange)
        rc = self._lib.qsv_create(
            int(n_qubits), _lib.QSV_F64 if dtype == "fp64" else _lib.QSV_F32, int(device), C.byref(cfg), C.byref(self._handle)
        )
        if rc != _lib.QSV_OK:
            msg = _lib.last_error(self._lib, None)
            if rc == _lib.QSV_E_ARG:
                raise ValueError(msg)
            raise CircuitEvaluatorException(f"qsv_create failed: {msg}")
        self._n_qubits = int(n_qubits)
        self._dtype = dtype
        self._group = max(1, int(self._lib.qsv_group_size(self._handle)))
        self._push_groups = 1  # launch groups per qsv_eval_push
        # (identities, circuits, ids, parameter counts, their sum) of the previous call: ONE tuple, replaced as a whole, so
        # that a thread never pairs one call's counts with another call's total (evaluators may share a device)
        self._last_batch = None
        self._row_counts = None
        self._ids_address = None
        self._push_evals = int(os.environ.get("QSV_PUSH_EVALS", "0"))  # measurement knob: evaluations per push
        self._push_plan = [int(x) for x in os.environ.get("QSV_PUSH_PLAN", "").split(",") if x]  # ... or explicit sizes
        self._operator: Optional[PauliOperator] = None
        self._reg_lock = threading.Lock()
        # key of this device in CircuitIR._registered: never reused, and unique across processes
        self._serial = (_PROCESS_TOKEN, next(_device_serial))
        # ids of circuits whose Python objects are gone.  Their finalizers only append here (they may run inside ANY
        # allocation, also between qsv_eval_begin and qsv_eval_end, where a call into the library would wait for the
        # handle this very thread holds); the ids are destroyed at the start of the next call that registers circuits.
        self._dead: list[int] = []
        self._dead_states: list[int] = []  # (ids of kept states whose KeptState objects are gone, likewise)
        self._watched: dict[int, tuple] = {}  # id(weak reference) -> (weak reference to a CircuitIR, its circuit id)
        self._gone = _make_gone(self._dead, self._watched)  # (holds the two containers, not the device)
        # held across "set the operator, then evaluate" by evaluators that share this device
        self.operator_lock = threading.RLock()

    # -- plumbing -------------------------------------------------------------------------------
    def __reduce__(self):
        return (_rebuild_device, (self._args, self._operator))

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def close(self) -> None:
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            getattr(self, "_watched", {}).clear()
            self._lib.qsv_destroy(handle)

    def _check(self, rc: int) -> None:
        if rc == _lib.QSV_OK:
            return
        msg = _lib.last_error(self._lib, self._handle)
        if rc == _lib.QSV_E_ARG:
            raise ValueError(msg)
        raise CircuitEvaluatorException(msg)

    @property
    def n_qubits(self) -> int:
        return self._n_qubits

    @property
    def device_index(self) -> int:
        """The HIP device this handle lives on."""
        return self._args[2]

    @property
    def dtype(self) -> str:
        return self._dtype

    def set_stream(self, hip_stream_ptr: int) -> None:
        """Launch on an existing HIP stream, e.g. ``torch.cuda.current_stream().cuda_stream``."""
        self._check(self._lib.qsv_set_stream(self._handle, C.c_void_p(hip_stream_ptr)))

    # -- operator -------------------------------------------------------------------------------
    def set_operator(self, operator: PauliOperator) -> None:
        if operator.num_qubits != self._n_qubits:
            raise ValueError(
                f"The operator acts on {operator.num_qubits} qubits but the device was created for {self._n_qubits}"
            )
        x = np.ascontiguousarray(operator.x_mask, dtype=np.uint64)
        z = np.ascontiguousarray(operator.z_mask, dtype=np.uint64)
        cre = np.ascontiguousarray(operator.coeffs.real, dtype=np.float64)
        cim = np.ascontiguousarray(operator.coeffs.imag, dtype=np.float64)
        self._check(
            self._lib.qsv_set_operator(
                self._handle, len(operator), _lib.as_ptr(x), _lib.as_ptr(z), _lib.as_ptr(cre), _lib.as_ptr(cim)
            )
        )
        self._operator = operator

    # -- circuits -------------------------------------------------------------------------------
    def circuit_id(self, circuit: CircuitIR) -> int:
        """Register ``circuit`` on this device once; later calls return the cached id."""
        cid = circuit._registered.get(self._serial)
        if cid is not None:
            return cid
        if circuit.n_qubits != self._n_qubits:
            raise ValueError(f"circuit has {circuit.n_qubits} qubits, the evaluator {self._n_qubits}")
        with self._reg_lock:
            self._reap()
            cid = circuit._registered.get(self._serial)
            if cid is None:
                ops = circuit.packed()
                out = C.c_int(0)
                kept = circuit._kept_state
                if kept is not None:
                    self._check(self._lib.qsv_circuit_create_on_prefix(self._handle, self._kept_id(kept), len(ops), _lib.as_ptr(ops),
                                                                       circuit.num_parameters, C.byref(out)))
                else:
                    self._check(
                        self._lib.qsv_circuit_create(self._handle, len(ops), _lib.as_ptr(ops), circuit.num_parameters, C.byref(out))
                    )
                cid = out.value
                circuit._registered[self._serial] = cid
                # drop the device-side plan when the circuit object goes away
                self._watch(circuit, cid)
        return cid

    def _register_many(self, fresh: Sequence[CircuitIR]) -> None:
        """Register several new circuit structures with ONE library call (``qsv_circuits_create``: the pass scheduler
        runs on several host threads).  A generation of EVQE brings up to a population of new structures at once."""
        fresh = list({id(c): c for c in fresh}.values())
        if len(fresh) < 2:
            return
        for c in fresh:
            if c.n_qubits != self._n_qubits:
                raise ValueError(f"circuit has {c.n_qubits} qubits, the evaluator {self._n_qubits}")
        with self._reg_lock:
            fresh = [c for c in fresh if self._serial not in c._registered]
            continued = [c for c in fresh if c._kept_state is not None]
            if continued:
                fresh = [c for c in fresh if c._kept_state is None]
                if len(continued) >= 2:
                    self._register_continued(continued)
            if len(fresh) < 2:
                return
            joined = b"".join([c._bytes for c in fresh])
            ops = np.frombuffer(joined, dtype=QSV_OP_DTYPE) if joined else np.zeros(1, dtype=QSV_OP_DTYPE)
            offsets = np.zeros(len(fresh) + 1, dtype=np.int64)
            np.cumsum([len(c._bytes) for c in fresh], out=offsets[1:])
            offsets //= QSV_OP_DTYPE.itemsize
            counts = np.asarray([c.num_parameters for c in fresh], dtype=np.int32)
            out = np.zeros(len(fresh), dtype=np.int32)
            self._check(self._lib.qsv_circuits_create(self._handle, len(fresh), _lib.as_ptr(offsets), _lib.as_ptr(ops),
                                                      _lib.as_ptr(counts), _lib.as_ptr(out)))
            for c, cid in zip(fresh, out.tolist()):
                c._registered[self._serial] = cid
                self._watch(c, cid)

    def _register_continued(self, circuits: Sequence[CircuitIR]) -> None:
        """``qsv_circuits_create_on_prefixes`` for circuits that continue kept states (caller holds ``_reg_lock``)."""
        joined = b"".join([c._bytes for c in circuits])
        ops = np.frombuffer(joined, dtype=QSV_OP_DTYPE) if joined else np.zeros(1, dtype=QSV_OP_DTYPE)
        offsets = np.zeros(len(circuits) + 1, dtype=np.int64)
        np.cumsum([len(c._bytes) for c in circuits], out=offsets[1:])
        offsets //= QSV_OP_DTYPE.itemsize
        counts = np.asarray([c.num_parameters for c in circuits], dtype=np.int32)
        states = np.asarray([self._kept_id(c._kept_state) for c in circuits], dtype=np.int32)
        out = np.zeros(len(circuits), dtype=np.int32)
        self._check(self._lib.qsv_circuits_create_on_prefixes(self._handle, len(circuits), _lib.as_ptr(offsets), _lib.as_ptr(ops),
                                                              _lib.as_ptr(counts), _lib.as_ptr(states), _lib.as_ptr(out)))
        for c, cid in zip(circuits, out.tolist()):
            c._registered[self._serial] = cid
            self._watch(c, cid)

    def _kept_id(self, kept: "KeptState") -> int:
        if kept._serial != self._serial or kept._id < 0:
            raise ValueError("the circuit continues a state that is not kept on this device (or was released)")
        return kept._id

    # -- kept states ------------------------------------------------------------------------------
    def keep_states(self, circuits: Sequence[CircuitIR], parameter_values: Sequence[Sequence[float]]) -> list[KeptState]:
        """Run every (circuit, parameter vector) pair from |0..0> once and keep its final state resident
        (``qsv_prefix_create``): circuits made with ``CircuitIR.continue_from(state)`` then start there.  What a layer search
        does with everything in front of the searched layer (reference: mutation.py:57-59)."""
        n = len(circuits)
        if len(parameter_values) != n:
            raise ValueError("circuits and parameter_values must have the same length")
        if n == 0:
            return []
        if any(c._kept_state is not None for c in circuits):
            raise ValueError("a kept state of a circuit that itself continues a kept state is not supported")
        self._register_many([c for c in circuits if self._serial not in c._registered])
        ids = np.fromiter((self.circuit_id(c) for c in circuits), dtype=np.int32, count=n)
        need = [c.num_parameters for c in circuits]
        counts = np.fromiter(map(len, parameter_values), dtype=np.int64, count=n)
        for i in range(n):
            if counts[i] < need[i]:
                raise ValueError(f"circuit {i} needs {need[i]} parameter values, got {int(counts[i])}")
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        flat = _pack_slice(parameter_values, 0, n, int(offsets[-1])) if offsets[-1] else np.zeros(1)
        out = np.zeros(n, dtype=np.int32)
        with self._reg_lock:
            self._reap()
        self._check(self._lib.qsv_prefix_create(self._handle, n, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(flat), _lib.as_ptr(out)))
        return [KeptState(self, pid) for pid in out.tolist()]

    def forget_last_batch(self) -> None:
        """Drop what the device remembers of the previous call (its circuit objects, by identity): circuits that continue kept
        states then die with their last user, and the states' memory is free for the next search."""
        self._last_batch = None
        self._row_counts = None
        self._ids_address = None

    def kept_state_count(self) -> int:
        """Kept states alive on the device (held by a ``KeptState`` or by a registered circuit)."""
        with self._reg_lock:
            self._reap()
        return int(self._lib.qsv_prefix_count(self._handle))

    def circuit_cost(self, circuit: CircuitIR) -> dict:
        """Which way an expectation value of ``circuit`` goes on this device under the operator set now, and about what it
        costs (``qsv_circuit_cost``): {"route", "n_keys", "n_passes", "on_kept_state", "microseconds"}."""
        cost = _lib.QsvCircuitCost()
        self._check(self._lib.qsv_circuit_cost(self._handle, self.circuit_id(circuit), C.byref(cost)))
        return {"route": _lib.ROUTE_NAMES[cost.route], "n_keys": cost.n_keys, "n_passes": cost.n_passes,
                "on_kept_state": bool(cost.on_kept_state), "microseconds": cost.microseconds}

    def _watch(self, circuit: CircuitIR, cid: int) -> None:
        """Note the device-side plan ``cid`` for destruction once ``circuit`` is garbage collected.  (A plain weak
        reference with a callback, kept alive in a dict: ``weakref.finalize`` cost 1.4 us per circuit -- 90 us of the
        registration of a generation's 64 new structures.)"""
        ref = weakref.ref(circuit, self._gone)  # (one callback per device, not one closure per circuit)
        self._watched[id(ref)] = (ref, cid)

    def _reap(self) -> None:
        """Destroy the device-side plans of circuits that were garbage collected (caller holds ``_reg_lock``)."""
        while self._dead:
            cid = self._dead.pop()
            if self._handle:
                self._lib.qsv_circuit_destroy(self._handle, cid)
        while self._dead_states:
            pid = C.c_int(self._dead_states.pop())
            if self._handle:
                self._lib.qsv_prefix_destroy(self._handle, 1, C.byref(pid))

    def _batch_metadata(self, circuits: Sequence[CircuitIR]) -> tuple[np.ndarray, np.ndarray, int]:
        """(circuit ids, parameter counts, sum of the counts) of a batch.  An optimiser calls with the same circuit objects over and
        over: both arrays are kept from the previous call, keyed by the objects' identities (the entry holds the
        circuits, so an identity cannot be recycled while it exists) and by the global edit counter."""
        n = len(circuits)
        cached = self._last_batch
        if cached is not None and cached[0][0] == CircuitIR.edits_of_registered:
            fast = _load_pyhelp_module()
            if fast is not None:
                if fast.same_objects(cached[1], circuits):  # (the entry holds the circuits: identities cannot be recycled)
                    return cached[2], cached[3], cached[4]
            elif cached[0] == (CircuitIR.edits_of_registered, *map(id, circuits)):
                return cached[2], cached[3], cached[4]
        if self._dead or self._dead_states:
            with self._reg_lock:
                self._reap()
        self._register_many([c for c in circuits if self._serial not in c._registered])
        ids = np.fromiter((self.circuit_id(c) for c in circuits), dtype=np.int32, count=n)
        need = np.fromiter((c.num_parameters for c in circuits), dtype=np.int64, count=n)
        total = int(need.sum())  # parameter values the batch takes
        self._last_batch = ((CircuitIR.edits_of_registered, *map(id, circuits)), list(circuits), ids, need, total)
        return ids, need, total

    def results_seen(self) -> None:
        """The caller has seen every result of the last batch that ended without waiting (``qsv_eval_results_seen``): the next
        call need not wait for the streams before it reuses the staging buffers."""
        self._check(self._lib.qsv_eval_results_seen(self._handle))

    def expectation_values_to_device(
        self, circuits: Sequence[CircuitIR], parameter_values: Sequence[Sequence[float]], device_pointer: int
    ) -> None:
        """:meth:`expectation_values` with the results left in DEVICE memory (``len(circuits)`` doubles at
        ``device_pointer``, this handle's GPU) and WITHOUT waiting for them (``qsv_eval_set_output``): they are complete
        once the work enqueued so far on the handle's stream (:meth:`set_stream`) is.  For a caller that runs something on
        that stream right behind -- the fitness all-gather of a sharded population (``queasars_amd.distributed``)."""
        n = len(circuits)
        if len(parameter_values) != n:
            raise ValueError("circuits and parameter_values must have the same length")
        if n == 0:
            return
        ids, need, total = self._batch_metadata(circuits)
        # (the helper is linked against the default libqsv.so: a handle of a diagnostic build, QSV_LIBRARY, is not its to touch)
        helper = None if os.environ.get("QSV_LIBRARY") else _load_pyhelp()
        if helper is not None:
            scratch = np.empty(total + 1, dtype=np.float64)
            self._check(helper.qsv_py_expectation_values_device(self._handle, n, ids.ctypes.data, need.ctypes.data,
                                                                parameter_values, scratch.ctypes.data, total,
                                                                C.c_void_p(device_pointer)))
            return
        counts = np.fromiter(map(len, parameter_values), dtype=np.int64, count=n)
        if (counts < need).any():
            i = int(np.argmax(counts < need))
            raise ValueError(f"circuit {i} needs {int(need[i])} parameter values, got {int(counts[i])}")
        lib, handle = self._lib, self._handle
        self._check(lib.qsv_eval_begin(handle, n, _lib.as_ptr(ids), _lib.as_ptr(counts)))
        rc = lib.qsv_eval_set_output(handle, C.c_void_p(device_pointer))
        if rc == _lib.QSV_OK:
            packed = _pack_slice(parameter_values, 0, n, int(counts.sum()))
            rc = lib.qsv_eval_push(handle, 0, n, _lib.as_ptr(packed))
        rc_end = lib.qsv_eval_end(handle, None)
        self._check(rc if rc != _lib.QSV_OK else rc_end)

    def expectation_values_of_device_parameters(
        self,
        circuits: Sequence[CircuitIR],
        device_pointer: int,
        row_length: int,
        ready_event: int = 0,
        out_device_pointer: int = 0,
    ) -> Optional[np.ndarray]:
        """:meth:`expectation_values` for parameter values that ALREADY LIVE IN DEVICE MEMORY (``qsv_eval_push_device``): a
        row-major ``len(circuits) x row_length`` matrix of doubles at ``device_pointer`` on this handle's GPU, circuit i
        taking the first ``num_parameters`` values of row i.  Nothing is packed and nothing crosses PCIe on the way in.
        ``ready_event``: a ``hipEvent_t`` (as an integer) after which the matrix is complete, 0 when it already is.  The matrix
        must stay unchanged until the call returns.  ``out_device_pointer``: as :meth:`expectation_values_to_device` (results
        left on the device, no wait, None returned)."""
        n = len(circuits)
        if n == 0:
            return None if out_device_pointer else np.zeros(0, dtype=np.float64)
        if row_length < 0 or (row_length > 0 and not device_pointer):
            raise ValueError("device_pointer / row_length do not describe a matrix")
        ids, need, _total = self._batch_metadata(circuits)
        cached = self._row_counts
        if cached is None or cached[0] != (n, row_length):
            counts = np.full(n, row_length, dtype=np.int64)
            cached = self._row_counts = ((n, row_length), counts, counts.ctypes.data)
        counts = cached[1]
        out = None if out_device_pointer else np.empty(n, dtype=np.float64)
        fast = None if os.environ.get("QSV_LIBRARY") else _load_pyhelp_module()
        if fast is not None:
            # (the whole batch in one call of the extension module; the address of the id array is kept with the array)
            kept = self._ids_address
            if kept is None or kept[0] is not ids:
                kept = self._ids_address = (ids, ids.ctypes.data)
            rc = fast.eval_device_matrix(self._handle.value, n, kept[1], cached[2], device_pointer, ready_event,
                                         out.ctypes.data if out is not None else 0, out_device_pointer)
            self._check(rc)
            return out
        helper = None if os.environ.get("QSV_LIBRARY") else _load_pyhelp()
        if helper is not None:
            self._check(helper.qsv_py_expectation_values_devparams(
                self._handle, n, ids.ctypes.data, counts.ctypes.data, device_pointer or None, ready_event or None,
                out.ctypes.data if out is not None else None, out_device_pointer or None))
            return out
        lib, handle = self._lib, self._handle
        self._check(lib.qsv_eval_begin(handle, n, _lib.as_ptr(ids), _lib.as_ptr(counts)))
        rc = lib.qsv_eval_set_output(handle, C.c_void_p(out_device_pointer)) if out_device_pointer else _lib.QSV_OK
        if rc == _lib.QSV_OK:
            rc = lib.qsv_eval_push_device(handle, 0, n, C.c_void_p(device_pointer) if row_length else None,
                                          C.c_void_p(ready_event) if ready_event else None)
        msg = _lib.last_error(lib, handle) if rc != _lib.QSV_OK else ""
        rc_end = lib.qsv_eval_end(handle, _lib.as_ptr(out) if out is not None else None)
        if rc != _lib.QSV_OK:
            raise (ValueError if rc == _lib.QSV_E_ARG else CircuitEvaluatorException)(msg)
        self._check(rc_end)
        return out

    def expectation_values(self, circuits: Sequence[CircuitIR], parameter_values: Sequence[Sequence[float]]) -> np.ndarray:
        """Exact ``real(<psi_i|H|psi_i>)`` for every (circuit, parameter vector) pair, in input order.

        Parameter vectors are converted to doubles a chunk at a time and pushed to the device as they
        become ready (``qsv_eval_begin / push / end``), so the conversion of the next group overlaps the GPU work
        on the previous one."""
        n = len(circuits)
        if len(parameter_values) != n:
            raise ValueError("circuits and parameter_values must have the same length")
        if n == 0:
            return np.zeros(0, dtype=np.float64)
        ids, need, total = self._batch_metadata(circuits)
        out = np.empty(n, dtype=np.float64)
        lib, handle = self._lib, self._handle
        helper = None if (self._push_evals or self._push_plan or os.environ.get("QSV_LIBRARY")) else _load_pyhelp()
        fast = _load_pyhelp_module() if helper is not None else None
        if fast is not None:
            # the whole begin / pack / push / end sequence in one call of the helper's extension module (csrc/pyhelp.c: no
            # ctypes argument conversion), which takes the first need[i] values of vector i and complains about a shorter one
            kept = self._ids_address
            if kept is None or kept[0] is not ids or len(kept) < 4:
                kept = self._ids_address = (ids, ids.ctypes.data, need, need.ctypes.data)
            scratch = np.empty(total + 1, dtype=np.float64)
            self._check(fast.eval_vectors(handle.value, n, kept[1], kept[3], parameter_values, scratch.ctypes.data, total,
                                          out.ctypes.data))
            return out
        if helper is not None:
            # (the same through ctypes.PyDLL)
            scratch = np.empty(total + 1, dtype=np.float64)
            rc = helper.qsv_py_expectation_values(handle, n, ids.ctypes.data, need.ctypes.data, parameter_values,
                                                  scratch.ctypes.data, total, out.ctypes.data)
            self._check(rc)
            return out
        counts = np.fromiter(map(len, parameter_values), dtype=np.int64, count=n)
        if (counts < need).any():
            i = int(np.argmax(counts < need))
            raise ValueError(f"circuit {i} needs {int(need[i])} parameter values, got {int(counts[i])}")
        self._check(lib.qsv_eval_begin(handle, n, _lib.as_ptr(ids), _lib.as_ptr(counts)))
        rc = _lib.QSV_OK
        try:
            # Two pushes per population: packing the second half overlaps the GPU work on the first, and the library
            # runs consecutive pushes on two HIP streams, so that the tail of one launch overlaps the next.  (While
            # packing cost 15 ns per value a small first push paid off; with the CPython-API packer it does not.)
            # Never more than a launch group per push.
            step = min(self._group * max(1, self._push_groups), max(8, (n + 1) // 2))
            if self._push_evals:
                step = self._push_evals
            bounds = list(range(0, n, step)) + [n]
            if not self._push_evals and not self._push_plan and 32 <= n <= self._group:
                bounds = [0, (n + 1) // 2, n]  # two halves, one per stream (measured: scripts/push_plan_sweep.sh)
            if self._push_plan:
                bounds, acc = [0], 0
                for size in self._push_plan:
                    acc = min(n, acc + size)
                    bounds.append(acc)
                while bounds[-1] < n:
                    bounds.append(min(n, bounds[-1] + self._push_plan[-1]))
            for first, last in zip(bounds[:-1], bounds[1:]):
                if last <= first:
                    continue
                total = int(counts[first:last].sum())
                if total:
                    values = _pack_slice(parameter_values, first, last, total)
                    rc = lib.qsv_eval_push(handle, first, last - first, _lib.as_ptr(values))
                else:
                    rc = lib.qsv_eval_push(handle, first, last - first, None)
                if rc != _lib.QSV_OK:
                    break
        finally:
            msg = _lib.last_error(lib, handle) if rc != _lib.QSV_OK else ""
            rc_end = lib.qsv_eval_end(handle, _lib.as_ptr(out))
        if rc != _lib.QSV_OK:
            raise (ValueError if rc == _lib.QSV_E_ARG else CircuitEvaluatorException)(msg)
        self._check(rc_end)
        return out

    def expectation_value_coalesced(self, circuit: CircuitIR, parameter_values: Sequence[float], window_us: float = 0.0) -> float:
        """One evaluation, merged inside the library with the evaluations other threads ask for at the same time
        (``qsv_eval_coalesced``).  The call blocks in C with the GIL released, so population_size Python threads calling
        with one circuit each (the reference's selection operator) are answered from one batch."""
        fast = _load_pyhelp_module()
        if fast is not None:
            # the whole call in one C function (csrc/pyhelp.c eval_one); None: the circuit is not registered here yet
            try:
                value = fast.eval_one(self._handle.value, self._serial, circuit, parameter_values, float(window_us))
                if value is None:
                    self.circuit_id(circuit)
                    value = fast.eval_one(self._handle.value, self._serial, circuit, parameter_values, float(window_us))
                return value
            except RuntimeError as exc:
                raise CircuitEvaluatorException(str(exc)) from None
        cid = circuit._registered.get(self._serial)
        if cid is None:
            cid = self.circuit_id(circuit)
        if len(parameter_values) < circuit.num_parameters:
            raise ValueError(f"circuit needs {circuit.num_parameters} parameter values, got {len(parameter_values)}")
        packed = array("d", parameter_values) if type(parameter_values) is list else array("d", list(parameter_values))
        out = C.c_double(0.0)
        address = packed.buffer_info()[0] if len(packed) else None
        rc = self._lib.qsv_eval_coalesced(self._handle, cid, address, len(packed), float(window_us), C.byref(out))
        if rc != _lib.QSV_OK:
            self._check(rc)
        return out.value

    def statevector(self, circuit: CircuitIR, parameter_values: Sequence[float]) -> np.ndarray:
        cid = self.circuit_id(circuit)
        p = np.ascontiguousarray(parameter_values, dtype=np.float64)
        out = np.empty(2 << self._n_qubits, dtype=np.float64)
        self._check(self._lib.qsv_statevector(self._handle, cid, _lib.as_ptr(p) if p.size else None, p.size, _lib.as_ptr(out)))
        return out.view(np.complex128)

    def probabilities(self, circuit: CircuitIR, parameter_values: Sequence[float]) -> np.ndarray:
        cid = self.circuit_id(circuit)
        p = np.ascontiguousarray(parameter_values, dtype=np.float64)
        out = np.empty(1 << self._n_qubits, dtype=np.float64)
        self._check(self._lib.qsv_probabilities(self._handle, cid, _lib.as_ptr(p) if p.size else None, p.size, _lib.as_ptr(out)))
        return out

    def sample(self, circuit: CircuitIR, parameter_values: Sequence[float], shots: int, seed: int) -> np.ndarray:
        return self.sample_batch([circuit], [parameter_values], shots, seed)[0][0]

    def sample_batch(
        self, circuits: Sequence[CircuitIR], parameter_values: Sequence[Sequence[float]], shots: int, seed: int, with_values: bool = False
    ) -> tuple[np.ndarray, Optional[np.ndarray]]:
        """``shots`` measured basis states per (circuit, parameter vector) pair, sampled on the device:
        ``states[i, s]``.  With ``with_values`` (diagonal operator set on the device) also ``values[i, s]``, the
        operator's value on each sample, gathered from the device-resident diagonal table."""
        n = len(circuits)
        if len(parameter_values) != n:
            raise ValueError("circuits and parameter_values must have the same length")
        states = np.empty((n, int(shots)), dtype=np.uint64)
        values = np.empty((n, int(shots)), dtype=np.float64) if with_values else None
        if n == 0 or shots == 0:
            return states, values
        ids, need, _ = self._batch_metadata(circuits)
        counts = np.fromiter(map(len, parameter_values), dtype=np.int64, count=n)
        if (counts < need).any():
            i = int(np.argmax(counts < need))
            raise ValueError(f"circuit {i} needs {int(need[i])} parameter values, got {int(counts[i])}")
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        flat = _pack_slice(parameter_values, 0, n, int(offsets[-1])) if offsets[-1] else np.zeros(1)
        self._check(
            self._lib.qsv_sample_batch(
                self._handle, n, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(flat), int(shots),
                C.c_uint64(seed & (2**64 - 1)), _lib.as_ptr(states), _lib.as_ptr(values) if with_values else None,
            )
        )
        return states, values

    #: most samples per evaluation the device-side CVaR sorts (csrc/kernels.hpp kCvarMaxShots)
    MAX_CVAR_SHOTS = 4096

    def sample_cvar_batch(
        self, circuits: Sequence[CircuitIR], parameter_values: Sequence[Sequence[float]], shots: int, seed: int, alpha: float
    ) -> list[float]:
        """CVaR_alpha of the (diagonal) operator over ``shots`` samples per (circuit, parameter vector) pair, sampled,
        valued and sorted on the device (``qsv_sample_cvar_batch``): the same samples :meth:`sample_batch` draws for
        ``seed``, but only one number per pair comes back."""
        n = len(circuits)
        if len(parameter_values) != n:
            raise ValueError("circuits and parameter_values must have the same length")
        if n == 0:
            return []
        ids, need, _ = self._batch_metadata(circuits)
        counts = np.fromiter(map(len, parameter_values), dtype=np.int64, count=n)
        if (counts < need).any():
            i = int(np.argmax(counts < need))
            raise ValueError(f"circuit {i} needs {int(need[i])} parameter values, got {int(counts[i])}")
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        flat = _pack_slice(parameter_values, 0, n, int(offsets[-1])) if offsets[-1] else np.zeros(1)
        out = np.empty(n, dtype=np.float64)
        self._check(
            self._lib.qsv_sample_cvar_batch(
                self._handle, n, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(flat), int(shots),
                C.c_uint64(seed & (2**64 - 1)), float(alpha), _lib.as_ptr(out),
            )
        )
        return out.tolist()

    def exact_cvar_batch(self, circuits: Sequence[CircuitIR], parameter_values: Sequence[Sequence[float]], alpha: float) -> list[float]:
        """CVaR_alpha of the (diagonal) operator under the EXACT output distribution of every (circuit, parameter vector)
        pair (``qsv_exact_cvar_batch``): what the reference's accumulation loop returns for a measured distribution that
        equals the exact one (expectation_calculation.py:14-32), stopping rule and tie order included.  Deterministic."""
        n = len(circuits)
        if len(parameter_values) != n:
            raise ValueError("circuits and parameter_values must have the same length")
        if n == 0:
            return []
        ids, need, _ = self._batch_metadata(circuits)
        counts = np.fromiter(map(len, parameter_values), dtype=np.int64, count=n)
        if (counts < need).any():
            i = int(np.argmax(counts < need))
            raise ValueError(f"circuit {i} needs {int(need[i])} parameter values, got {int(counts[i])}")
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        flat = _pack_slice(parameter_values, 0, n, int(offsets[-1])) if offsets[-1] else np.zeros(1)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.qsv_exact_cvar_batch(self._handle, n, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(flat),
                                                   float(alpha), _lib.as_ptr(out)))
        return out.tolist()

    # -- measurement support ----------------------------------------------------------------------
    def set_option(self, name: str, value: int) -> None:
        """Switches of the handle (``qsv_set_option``): "split", "factor", "split_sampling" (0 / 1), "streams" (1 .. 4).
        A circuit keeps the form it was registered in; the cache of the previous batch is dropped."""
        self._check(self._lib.qsv_set_option(self._handle, name.encode(), int(value)))
        self._last_batch = None
        self._row_counts = None
        self._ids_address = None

    def set_profiling(self, enabled: bool) -> None:
        self._check(self._lib.qsv_set_profiling(self._handle, 1 if enabled else 0))

    def profile(self) -> dict:
        prof = _lib.QsvProfile()
        self._check(self._lib.qsv_get_profile(self._handle, C.byref(prof)))
        out = {}
        for name, ctype in prof._fields_:
            value = getattr(prof, name)
            out[name] = list(value) if hasattr(value, "__len__") else value
        return out

    def bench_gate(self, target: int, control: int = -1, theta=1.0, phi=0.5, lam=0.25, reps: int = 100) -> float:
        """Average device milliseconds of one read-modify-write sweep applying a single u / cu3 gate."""
        ms = C.c_double(0.0)
        self._check(self._lib.qsv_bench_gate(self._handle, target, control, theta, phi, lam, reps, C.byref(ms)))
        return ms.value

    def bench_ops(self, circuit: CircuitIR, reps: int = 20) -> tuple[float, int]:
        """(milliseconds per repetition, passes per repetition) of a bound circuit applied read-modify-write."""
        ops = circuit.packed()
        ms, n_passes = C.c_double(0.0), C.c_int(0)
        self._check(self._lib.qsv_bench_ops(self._handle, len(ops), _lib.as_ptr(ops), reps, C.byref(ms), C.byref(n_passes)))
        return ms.value, n_passes.value


def _rebuild_device(args, operator):
    n_qubits, dtype, device, tile_bits, reg_bits, low_bits, group, exchange = args
    dev = StatevectorDevice(n_qubits, dtype, device, tile_bits, reg_bits, low_bits, group, exchange)
    if operator is not None:
        dev.set_operator(operator)
    return dev


class _ComposedCircuits:
    """``initial_state_circuit + circuit`` for the circuits an evaluator is called with, without pinning them: entries
    are keyed by identity, hold the user's circuit only weakly (the entry goes when the circuit goes, and with it the
    composed circuit and its device-side plan), notice in-place edits through the circuit's version counter, and are
    bounded in number (the reference creates fresh circuits on every call: such entries are dead weight)."""

    def __init__(self, initial_state_circuit: Optional[CircuitIR], limit: int = 1024):
        self._initial = initial_state_circuit
        self._limit = int(limit)
        self._entries: dict[int, tuple[weakref.ref, int, CircuitIR]] = {}
        self._lock = threading.Lock()

    def __len__(self) -> int:
        return len(self._entries)

    def __reduce__(self):  # a cache travels empty (evaluators are pickled to process-based executors)
        return (_ComposedCircuits, (self._initial, self._limit))

    def get(self, circuit: CircuitIR) -> CircuitIR:
        if self._initial is None or circuit._kept_state is not None:  # (a kept state already has the initial state in it)
            return circuit
        key = id(circuit)
        hit = self._entries.get(key)
        if hit is not None and hit[0]() is circuit and hit[1] == circuit._version:
            return hit[2]
        composed = self._initial.compose(circuit)
        with self._lock:
            if len(self._entries) >= self._limit:
                for old in list(self._entries)[: self._limit // 2]:  # dicts keep insertion order: drop the oldest half
                    self._entries.pop(old, None)
            entries = self._entries
            self._entries[key] = (weakref.ref(circuit, lambda _r, k=key: entries.pop(k, None)), circuit._version, composed)
        return composed


def _check_initial_state(initial_state_circuit: Optional[CircuitIR], n_qubits: int, what: str) -> None:
    if initial_state_circuit is not None and initial_state_circuit.num_qubits != n_qubits:
        raise ValueError(
            f"The amount of qubits in the initial state circuit ({initial_state_circuit.num_qubits} "
            + f"does not match {what} ({n_qubits})"
        )


class OperatorCircuitEvaluator(BaseCircuitEvaluator):
    """Exact expectation values of ``operator`` on the GPU (estimator branch of the reference).

    :param operator: observable; if it is not hermitian the imaginary part of the result is dropped
    :param estimator_precision: standard deviation of optional Gaussian noise added on the host to the exact
        value, seeded by ``seed`` (the reference's estimators emulate shot noise this way; 0 = exact)
    :param initial_state_circuit: optional circuit prepended to every evaluated circuit; it must not have free
        parameters and must act on exactly as many qubits as the operator
    """

    def __init__(
        self,
        operator: PauliOperator,
        estimator_precision: float = 0.0,
        initial_state_circuit: Optional[CircuitIR] = None,
        dtype: str = "fp64",
        device: int = 0,
        seed: Optional[int] = None,
        statevector_device: Optional[StatevectorDevice] = None,
    ):
        if not isinstance(operator, PauliOperator):
            raise ValueError("The operator must be a PauliOperator!")
        if estimator_precision < 0:
            raise ValueError("estimator_precision must not be negative!")
        _check_initial_state(initial_state_circuit, operator.num_qubits, "the amount of qubits in the given operator")
        if initial_state_circuit is not None and initial_state_circuit.num_parameters:
            raise ValueError("The initial state circuit must not have free parameters!")
        self._operator = operator
        self._precision = float(estimator_precision)
        self._initial_state_circuit = initial_state_circuit
        self._rng = np.random.default_rng(seed)
        self._device = statevector_device or StatevectorDevice(operator.num_qubits, dtype=dtype, device=device)
        if self._device.n_qubits != operator.num_qubits:
            raise ValueError("statevector_device was created for a different number of qubits")
        with self._device.operator_lock:
            self._device.set_operator(operator)
        self._composed = _ComposedCircuits(initial_state_circuit)
        self._composed_lists = None
        self._last_matrix = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_last_matrix"] = None  # (a weak reference to a tensor of this process)
        return state

    def _with_initial_state(self, circuit: CircuitIR) -> CircuitIR:
        return self._composed.get(circuit)

    def evaluate_circuits(self, circuits: list[CircuitIR], parameter_values: list[list[float]]) -> list[float]:
        """``parameter_values`` may also be a 2-D float64 tensor in THIS device's memory (anything with ``is_cuda`` /
        ``data_ptr()``, i.e. a ``torch.Tensor``; one row per circuit, a circuit takes the first ``num_parameters`` values of its
        row): the kernels then read the values where they are (``qsv_eval_push_device``), after the work queued so far on the
        tensor's current stream.  (A tensor that something torch does not see writes to -- another library, through its
        pointer -- has to be complete when it is handed over.)"""
        matrix = parameter_values if getattr(parameter_values, "is_cuda", False) else None
        if matrix is None and (_has_none(circuits) or _has_none(parameter_values)):
            pairs = [(c, p) for c, p in zip(circuits, parameter_values) if c is not None and p is not None]
            circuits, parameter_values = [c for c, _ in pairs], [p for _, p in pairs]
        if self._initial_state_circuit is not None:
            circuits = [self._with_initial_state(c) for c in circuits]
        # evaluators may share one device: "is it my operator? else set it" and the evaluation are one critical section
        with self._device.operator_lock:
            if self._device._operator is not self._operator:
                self._device.set_operator(self._operator)
            if matrix is not None:
                values = self._evaluate_device_matrix(circuits, matrix)
            else:
                values = self._device.expectation_values(circuits, parameter_values)
        if self._precision > 0:
            values = values + self._rng.normal(0.0, self._precision, size=values.shape)
        return values.tolist()

    def _evaluate_device_matrix(self, circuits, matrix, ready: bool = False, out_device_pointer: int = 0) -> Optional[np.ndarray]:
        import torch

        # (the matrix this evaluator read last, untouched by any torch operation since -- same storage owner, same version
        # counter, same place and shape: an optimiser's population evaluated again, a benchmark's resident input -- is as
        # complete as it was then, and as well-formed)
        base = matrix._base
        owner = base if base is not None else matrix
        shape = matrix.shape
        stamp = (matrix._version, matrix.data_ptr(), shape[0], shape[1] if len(shape) == 2 else -1)
        last = self._last_matrix
        seen = last is not None and last[0]() is owner and last[1] == stamp
        if not seen:
            if matrix.dim() != 2 or matrix.dtype != torch.float64 or not matrix.is_contiguous():
                raise ValueError("a device-resident parameter matrix must be a contiguous 2-D float64 tensor")
            if matrix.device.index != self._device.device_index:
                raise ValueError("the parameter matrix lives on another device than the evaluator")
        if shape[0] != len(circuits):
            raise ValueError("circuits and parameter_values must have the same length")
        if _has_none(circuits):
            raise ValueError("a device-resident parameter matrix cannot skip circuits (None entries)")
        event = 0
        if seen:
            ready = True
        stream = None if ready else torch.cuda.current_stream(matrix.device)
        if stream is not None and not stream.query():
            # (whatever produces the matrix was queued on the tensor's current stream and has not finished: the handle's
            # streams wait for it.  An idle stream -- the usual case -- costs one query instead of an event and four waits.)
            # (an event of this call's own: evaluators are shared by threads, whose tensors may come from different streams)
            marker = torch.cuda.Event()
            marker.record(stream)
            event = marker.cuda_event
        out = self._device.expectation_values_of_device_parameters(circuits, stamp[1], stamp[3], event, out_device_pointer)
        if not seen:
            self._last_matrix = (weakref.ref(owner), stamp)
        return out

    def evaluate_device_parameters(self, circuits: list[CircuitIR], matrix, ready: bool = False) -> np.ndarray:
        """:meth:`evaluate_circuits` for a device-resident parameter matrix, as a NumPy array.  ``ready=True``: the matrix is
        complete already (the caller synchronised, or it has not changed since an earlier call): no event is recorded."""
        if self._precision > 0:
            raise ValueError("estimator_precision > 0 is emulated on the host: use evaluate_circuits")
        if self._initial_state_circuit is not None:
            circuits = [self._with_initial_state(c) for c in circuits]
        with self._device.operator_lock:
            if self._device._operator is not self._operator:
                self._device.set_operator(self._operator)
            return self._evaluate_device_matrix(circuits, matrix, ready)

    def keep_states(self, circuits: list[CircuitIR], parameter_values: list[list[float]]) -> list[KeptState]:
        """The final states of the (circuit, parameter vector) pairs -- behind this evaluator's initial state, if it has one --
        kept resident on the device (:meth:`StatevectorDevice.keep_states`); circuits made with
        ``CircuitIR.continue_from(state)`` are then evaluated from there by every method of this class."""
        if self._initial_state_circuit is not None:
            circuits = [self._with_initial_state(c) for c in circuits]
        return self._device.keep_states(circuits, parameter_values)

    def forget_circuits(self) -> None:
        """:meth:`StatevectorDevice.forget_last_batch`, and this evaluator's own memory of its last list of circuits."""
        self._composed_lists = None
        self._device.forget_last_batch()

    def circuit_costs(self, circuits: list[CircuitIR]) -> list[dict]:
        """:meth:`StatevectorDevice.circuit_cost` of every circuit as this evaluator would run it (its operator set)."""
        if self._initial_state_circuit is not None:
            circuits = [self._with_initial_state(c) for c in circuits]
        with self._device.operator_lock:
            if self._device._operator is not self._operator:
                self._device.set_operator(self._operator)
            self._device._register_many([c for c in circuits if self._device._serial not in c._registered])
            return [self._device.circuit_cost(c) for c in circuits]

    def device_resident_search_possible(self) -> bool:
        """Can an optimiser keep its points and values in this evaluator's device memory (:meth:`evaluate_device_to_device`)?
        Only the exact estimator: noise is emulated on the host."""
        if self._precision > 0:
            return False
        try:
            import torch
        except ImportError:
            return False
        return torch.cuda.is_available()

    def evaluate_device_to_device(self, circuits: list[CircuitIR], matrix, out) -> None:
        """Parameter values from a device matrix (one row per circuit), expectation values into the device tensor ``out``
        (``len(circuits)`` doubles), nothing waited for and nothing copied: both tensors belong to the stream the evaluator's
        handle launches on (``StatevectorDevice.set_stream``) -- work queued there before the call is seen, work queued after
        it sees the values.  ``circuits`` should be the same list object call after call (its composition with an initial
        state and its device-side ids are kept by identity)."""
        if self._precision > 0:
            raise ValueError("estimator_precision > 0 is emulated on the host")
        if self._initial_state_circuit is not None:
            kept = self._composed_lists
            if kept is None or kept[0] is not circuits:
                kept = self._composed_lists = (circuits, [self._with_initial_state(c) for c in circuits])
            circuits = kept[1]
        with self._device.operator_lock:
            if self._device._operator is not self._operator:
                self._device.set_operator(self._operator)
            self._evaluate_device_matrix(circuits, matrix, ready=True, out_device_pointer=out.data_ptr())

    def evaluate_circuits_to_device(self, circuits: list[CircuitIR], parameter_values: list[list[float]], device_pointer: int) -> bool:
        """:meth:`evaluate_circuits` with the values left in device memory and without waiting for them
        (:meth:`StatevectorDevice.expectation_values_to_device`).  Only the exact estimator without missing entries can do
        that; returns False -- nothing was started -- otherwise."""
        matrix = parameter_values if getattr(parameter_values, "is_cuda", False) else None
        if self._precision > 0 or _has_none(circuits) or (matrix is None and _has_none(parameter_values)):
            return False
        if self._initial_state_circuit is not None:
            circuits = [self._with_initial_state(c) for c in circuits]
        with self._device.operator_lock:
            if self._device._operator is not self._operator:
                self._device.set_operator(self._operator)
            if matrix is not None:
                self._evaluate_device_matrix(circuits, matrix, out_device_pointer=device_pointer)
            else:
                self._device.expectation_values_to_device(circuits, parameter_values, device_pointer)
        return True

    @property
    def n_qubits(self) -> int:
        return self._operator.num_qubits

    @property
    def statevector_device(self) -> StatevectorDevice:
        return self._device


def measure_quasi_distributions(
    circuits: list[CircuitIR],
    parameter_values: list[list[float]],
    sampler: StatevectorDevice,
    shots: int,
    seed: Optional[int] = None,
) -> list[dict[int, float]]:
    """``{state: count / shots}`` per circuit, sampled on the device in one batched call (reference [29-59])."""
    pairs = [(c, p) for c, p in zip(circuits, parameter_values) if c is not None and p is not None]
    rng = np.random.default_rng(seed)
    states, _ = sampler.sample_batch([c for c, _ in pairs], [p for _, p in pairs], shots, int(rng.integers(0, 2**63 - 1)))
    out = []
    for row in states:
        values, counts = np.unique(row, return_counts=True)
        out.append({int(s): int(c) / shots for s, c in zip(values, counts)})
    return out


def _cvar_of_samples(values: np.ndarray, alpha: float) -> float:
    """Expectation / CVaR_alpha of equally weighted samples: what `_get_expectation` computes on the measured
    distribution (reference: expectation_calculation.py:14-32), evaluated on the sorted sample values."""
    shots = values.size
    if np.isclose(alpha, 1):
        return float(values.mean())
    ordered = np.sort(values)
    # gather probability mass alpha in ascending order of value: whole samples, then a fraction of the next one
    mass = alpha * shots
    whole = int(np.floor(mass + 1e-12))
    total = float(ordered[:whole].sum())
    if whole < shots and mass - whole > 1e-12:
        total += (mass - whole) * float(ordered[whole])
    return total / mass


def _cvar_of_sample_matrix(values: np.ndarray, alpha: float) -> list[float]:
    """:func:`_cvar_of_samples` for every row of ``values`` (one row of ``shots`` sample values per circuit) at once: one
    sort of the whole matrix instead of one NumPy call chain per circuit (5 us each: as much as the device took to
    produce the samples)."""
    if values.size == 0:
        return []
    shots = values.shape[1]
    if np.isclose(alpha, 1):
        return values.mean(axis=1).tolist()
    ordered = np.sort(values, axis=1)
    mass = alpha * shots
    whole = int(np.floor(mass + 1e-12))
    total = ordered[:, :whole].sum(axis=1)
    if whole < shots and mass - whole > 1e-12:
        total = total + (mass - whole) * ordered[:, whole]
    return (total / mass).tolist()


class OperatorSamplerCircuitEvaluator(BaseCircuitEvaluator):
    """Expectation / CVaR_alpha of a diagonal operator from ``sampler_shots`` measurements (reference [94-161]).

    ``sampler_shots=None`` (not in the reference, whose samplers always draw): the same quantity for the EXACT output
    distribution -- no sampling noise, deterministic, the limit the sampled values scatter around -- computed on the
    device (:meth:`StatevectorDevice.exact_cvar_batch`)."""

    def __init__(
        self,
        sampler_shots: Optional[int],
        operator: PauliOperator,
        alpha: float = 1.0,
        initial_state_circuit: Optional[CircuitIR] = None,
        dtype: str = "fp64",
        device: int = 0,
        seed: Optional[int] = None,
        statevector_device: Optional[StatevectorDevice] = None,
    ):
        if not isinstance(operator, PauliOperator):
            raise ValueError(
                "If using a sampler to estimate the expectation value, the operator must be a SparsePauliOp!"
            )
        if not operator.is_diagonal():
            raise ValueError("The sampler branch needs a diagonal (I/Z only) operator!")
        if alpha <= 0 or 1 < alpha:
            raise ValueError("alpha must be in the range (0, 1]!")
        _check_initial_state(initial_state_circuit, operator.num_qubits, "the amount of qubits in the given operator")
        self._operator = operator
        if sampler_shots is not None and int(sampler_shots) < 1:
            raise ValueError("sampler_shots must be a positive number of shots, or None for the exact distribution!")
        self._shots = None if sampler_shots is None else int(sampler_shots)
        self._alpha = float(alpha)
        self._initial_state_circuit = initial_state_circuit
        self._rng = np.random.default_rng(seed)
        self._device = statevector_device or StatevectorDevice(operator.num_qubits, dtype=dtype, device=device)
        with self._device.operator_lock:
            self._device.set_operator(operator)
        self._composed = _ComposedCircuits(initial_state_circuit)

    @property
    def statevector_device(self) -> StatevectorDevice:
        return self._device

    def evaluate_circuits(self, circuits: list[CircuitIR], parameter_values: list[list[float]]) -> list[float]:
        """Samples every circuit on the device, gathers each sample's operator value from the device-resident diagonal
        table and takes the CVaR there as well (up to 4096 shots; beyond that the host sorts the values)."""
        pairs = [(self._composed.get(c), p) for c, p in zip(circuits, parameter_values) if c is not None and p is not None]
        seed = int(self._rng.integers(0, 2**63 - 1))
        with self._device.operator_lock:
            if self._device._operator is not self._operator:
                self._device.set_operator(self._operator)
            if self._shots is None:
                if np.isclose(self._alpha, 1):  # (the reference takes the plain mean there: the expectation value)
                    return self._device.expectation_values([c for c, _ in pairs], [p for _, p in pairs]).tolist()
                return self._device.exact_cvar_batch([c for c, _ in pairs], [p for _, p in pairs], self._alpha)
            if self._shots <= StatevectorDevice.MAX_CVAR_SHOTS:
                return self._device.sample_cvar_batch([c for c, _ in pairs], [p for _, p in pairs], self._shots, seed, self._alpha)
            _, values = self._device.sample_batch([c for c, _ in pairs], [p for _, p in pairs], self._shots, seed, with_values=True)
        return _cvar_of_sample_matrix(values, self._alpha)

    @property
    def n_qubits(self) -> int:
        return self._operator.num_qubits


class BitstringCircuitEvaluator(BaseCircuitEvaluator):
    """Expectation / CVaR of a host-side bitstring scoring function over sampled measurements (reference [222-291]).
    The scoring callable stays on the host; the GPU supplies the samples."""

    def __init__(
        self,
        sampler_shots: int,
        bitstring_evaluator: BitstringEvaluator,
        alpha: float = 1.0,
        initial_state_circuit: Optional[CircuitIR] = None,
        dtype: str = "fp64",
        device: int = 0,
        seed: Optional[int] = None,
        statevector_device: Optional[StatevectorDevice] = None,
    ):
        _check_initial_state(
            initial_state_circuit, bitstring_evaluator.input_length, "the input length of the BitstringEvaluator"
        )
        if alpha <= 0 or 1 < alpha:
            raise ValueError("alpha must be in the range (0, 1]!")
        self._bitstring_evaluator = bitstring_evaluator
        self._shots = int(sampler_shots)
        self._alpha = float(alpha)
        self._initial_state_circuit = initial_state_circuit
        self._rng = np.random.default_rng(seed)
        self._device = statevector_device or StatevectorDevice(bitstring_evaluator.input_length, dtype=dtype, device=device)
        self._composed = _ComposedCircuits(initial_state_circuit)

    def evaluate_circuits(self, circuits: list[CircuitIR], parameter_values: list[list[float]]) -> list[float]:
        circuits = [None if c is None else self._composed.get(c) for c in circuits]
        dists = measure_quasi_distributions(
            circuits, parameter_values, self._device, self._shots, seed=int(self._rng.integers(0, 2**63 - 1))
        )
        return [
            get_expectation_with_bitstring_evaluator(d, self._bitstring_evaluator, self._alpha, self.n_qubits) for d in dists
        ]

    @property
    def n_qubits(self) -> int:
        return self._bitstring_evaluator.input_length
