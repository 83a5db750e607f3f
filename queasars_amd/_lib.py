"""ctypes binding of libqsv (include/qsv.h).  No CPU fallback: a missing or unloadable library raises."""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

import numpy as np

from queasars_amd import _build

QSV_OK, QSV_E_ARG, QSV_E_DEVICE, QSV_E_STATE, QSV_E_UNSUPPORTED = 0, -1, -2, -3, -4
QSV_F64, QSV_F32 = 0, 1


class QsvOp(C.Structure):
    _fields_ = [
        ("kind", C.c_uint8),
        ("target", C.c_uint8),
        ("control", C.c_uint8),
        ("flags", C.c_uint8),
        ("p_theta", C.c_int32),
        ("p_phi", C.c_int32),
        ("p_lambda", C.c_int32),
        ("theta", C.c_double),
        ("phi", C.c_double),
        ("lam", C.c_double),
    ]


class QsvPlanConfig(C.Structure):
    _fields_ = [
        ("tile_bits", C.c_int32),
        ("reg_bits", C.c_int32),
        ("low_bits", C.c_int32),
        ("group", C.c_int32),
        ("exchange", C.c_int32),
    ]


class QsvSpsaStepArgs(C.Structure):
    """``qsv_spsa_step_args`` of include/qsv.h (device pointers as integers)."""

    _fields_ = [
        ("n_runs", C.c_int32),
        ("width", C.c_int32),
        ("x", C.c_void_p),
        ("active", C.c_void_p),
        ("iterations", C.c_void_p),
        ("delta_accept", C.c_void_p),
        ("values", C.c_void_p),
        ("delta_propose", C.c_void_p),
        ("points", C.c_void_p),
        ("eps", C.c_double),
        ("lr", C.c_double),
        ("trust_region", C.c_int32),
        ("maxiter", C.c_int32),
        ("window", C.c_int32),
        ("reserved", C.c_int32),
        ("min_rel", C.c_double),
        ("maxfev", C.c_int64),
        ("previous", C.c_void_p),
        ("n_values", C.c_void_p),
        ("changes", C.c_void_p),
    ]


class QsvProfile(C.Structure):
    _fields_ = [
        ("n_evals", C.c_uint64),
        ("n_pass_launches", C.c_uint64),
        ("n_state_passes", C.c_uint64),
        ("n_gates", C.c_uint64),
        ("state_bytes", C.c_uint64),
        ("pass_ms", C.c_double),
        ("expect_ms", C.c_double),
        ("total_ms", C.c_double),
        ("pass_window_ms", C.c_double),
        ("moved_bytes", C.c_uint64),
        ("kernel_launches", C.c_uint64 * 3),
        ("kernel_ms", C.c_double * 3),
        ("kernel_bytes", C.c_uint64 * 3),
        ("kernel_moved_bytes", C.c_uint64 * 3),
        ("kernel_flops", C.c_double * 3),
        ("kernel_states", C.c_uint64 * 3),
    ]


class QsvCircuitCost(C.Structure):
    """``qsv_circuit_cost_t`` of include/qsv.h."""

    _fields_ = [
        ("route", C.c_int32),
        ("n_keys", C.c_int32),
        ("n_passes", C.c_int32),
        ("on_kept_state", C.c_int32),
        ("microseconds", C.c_double),
    ]


ROUTE_NAMES = ("one tile", "split, one launch", "split", "gate passes")

assert C.sizeof(QsvOp) == 40

# every symbol include/qsv.h declares: (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "qsv_version": (C.c_char_p, []),
    "qsv_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(QsvPlanConfig), C.POINTER(_P)]),
    "qsv_destroy": (None, [_P]),
    "qsv_last_error": (C.c_char_p, [_P]),
    "qsv_set_stream": (C.c_int, [_P, _P]),
    "qsv_n_qubits": (C.c_int, [_P]),
    "qsv_set_operator": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "qsv_circuit_create": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "qsv_circuits_create": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "qsv_circuit_destroy": (C.c_int, [_P, C.c_int]),
    "qsv_prefix_create": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "qsv_prefix_destroy": (C.c_int, [_P, C.c_int, _P]),
    "qsv_prefix_count": (C.c_int, [_P]),
    "qsv_circuit_create_on_prefix": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "qsv_circuits_create_on_prefixes": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "qsv_circuit_cost": (C.c_int, [_P, C.c_int, C.POINTER(QsvCircuitCost)]),
    "qsv_eval_circuits": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "qsv_eval_coalesced": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "qsv_eval_begin": (C.c_int, [_P, C.c_int, _P, _P]),
    "qsv_eval_push": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "qsv_eval_staging": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "qsv_eval_push_device": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "qsv_spsa_step": (C.c_int, [_P, _P]),
    "qsv_eval_end": (C.c_int, [_P, _P]),
    "qsv_eval_set_output": (C.c_int, [_P, _P]),
    "qsv_eval_results_seen": (C.c_int, [_P]),
    "qsv_eval_suggested_pushes": (C.c_int, [_P]),
    "qsv_group_size": (C.c_int, [_P]),
    "qsv_eval_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "qsv_statevector": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "qsv_probabilities": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "qsv_sample": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_uint64, _P]),
    "qsv_sample_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_uint64, _P, _P]),
    "qsv_sample_cvar_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_uint64, C.c_double, _P]),
    "qsv_exact_cvar_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_double, _P]),
    "qsv_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "qsv_fitness_table_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int]),
    "qsv_set_profiling": (C.c_int, [_P, C.c_int]),
    "qsv_get_profile": (C.c_int, [_P, C.POINTER(QsvProfile)]),
    "qsv_bench_gate": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "qsv_bench_ops": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "qsv_split_describe": (
        C.c_int,
        [C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_uint64), _P, C.c_int, C.POINTER(C.c_int), _P, C.c_int, C.POINTER(C.c_int)],
    ),
    "qsv_plan_build": (
        C.c_int,
        [C.c_int, C.c_int, C.c_int, _P, C.POINTER(QsvPlanConfig), _P, C.c_size_t, C.POINTER(C.c_size_t)],
    ),
}

_lock = threading.Lock()
_lib = None


class QsvLibraryError(RuntimeError):
    """libqsv is missing or could not be loaded.  There is deliberately no fallback."""


def library_path() -> Path:
    """queasars_amd/libqsv.so; QSV_LIBRARY points measurement scripts at a diagnostic build of the same sources."""
    override = os.environ.get("QSV_LIBRARY")
    return Path(override) if override else _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load queasars_amd/libqsv.so (building it first when absent and hipcc is available)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not path.exists():
            if not build_if_missing:
                raise QsvLibraryError(f"{path} is missing; run `python -c 'import __graft_entry__ as g; g.build()'`")
            _build.build()
        elif build_if_missing and path == _build.LIB_PATH and _build.needs_build() and _build.have_hipcc():
            _build.build()  # a source is newer than the library: never run a stale build silently
        if os.environ.get("QSV_NO_TORCH") != "1":
            # torch ships its own HIP runtime (same soname as /opt/rocm's).  Importing it first makes libqsv and
            # torch share ONE runtime instance, so streams and events are interchangeable between them.
            import torch  # noqa: F401
        try:
            lib = C.CDLL(str(path))
        except OSError as exc:  # pragma: no cover - depends on the machine
            raise QsvLibraryError(f"cannot load {path}: {exc}") from exc
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise QsvLibraryError(f"{path} does not export {name}") from exc
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib


def as_ptr(arr: np.ndarray) -> C.c_void_p:
    return C.c_void_p(arr.ctypes.data)


def last_error(lib: C.CDLL, handle) -> str:
    msg = lib.qsv_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""
