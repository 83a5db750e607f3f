"""Host-only access to libqsv's pass scheduler (no device needed): used by tests and by tuning scripts."""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from queasars_amd import _lib
from queasars_amd.ir import CircuitIR


def build_plan_words(
    circuit: CircuitIR,
    dtype: int = _lib.QSV_F64,
    tile_bits: int = 0,
    reg_bits: int = 0,
    low_bits: int = 0,
    exchange: int = 0,
) -> np.ndarray:
    """Encoded pass plan (uint32 words) the scheduler produces for ``circuit``."""
    lib = _lib.load()
    ops = circuit.packed()
    cfg = _lib.QsvPlanConfig(tile_bits, reg_bits, low_bits, 0, exchange)
    n_words = C.c_size_t(0)
    rc = lib.qsv_plan_build(circuit.n_qubits, dtype, len(ops), _lib.as_ptr(ops), C.byref(cfg), None, 0, C.byref(n_words))
    if rc != 0:
        raise ValueError(f"qsv_plan_build failed ({rc}): {_lib.last_error(lib, None)}")
    words = np.zeros(n_words.value, dtype=np.uint32)
    rc = lib.qsv_plan_build(
        circuit.n_qubits, dtype, len(ops), _lib.as_ptr(ops), C.byref(cfg), _lib.as_ptr(words), words.size, C.byref(n_words)
    )
    if rc != 0:
        raise ValueError(f"qsv_plan_build failed ({rc}): {_lib.last_error(lib, None)}")
    return words


def gate_matrices(circuit: CircuitIR, parameter_values) -> np.ndarray:
    """(n_gates, 8) array of the non-identity gates' matrices, m00 m01 m10 m11 as (re, im) pairs."""
    import math

    rows = []
    for kind, _t, _c, theta, phi, lam in circuit.bound_ops(parameter_values):
        if kind == 0:
            continue
        c, s = math.cos(theta / 2), math.sin(theta / 2)
        rows.append(
            [c, 0.0, -math.cos(lam) * s, -math.sin(lam) * s, math.cos(phi) * s, math.sin(phi) * s,
             math.cos(phi + lam) * c, math.sin(phi + lam) * c]
        )
    return np.asarray(rows, dtype=np.float64).reshape(-1, 8)
