"""The C-ABI library loads without a GPU, exports every symbol include/qsv.h declares, and fails loudly (no
fallback) when asked to compute without a device."""

import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from queasars_amd import _lib
from queasars_amd.ir import QSV_OP_DTYPE, CircuitIR, PauliOperator

ROOT = Path(__file__).resolve().parent.parent


def declared_functions() -> list[str]:
    text = (ROOT / "include" / "qsv.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qsv_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = declared_functions()
    assert "qsv_eval_circuits" in declared and "qsv_create" in declared
    assert sorted(_lib.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert b"libqsv" in lib.qsv_version()


def test_struct_layouts_match_the_header():
    assert QSV_OP_DTYPE.itemsize == C.sizeof(_lib.QsvOp) == 40
    for field, offset in (("kind", 0), ("target", 1), ("control", 2), ("p_theta", 4), ("p_lambda", 12), ("theta", 16), ("lam", 32)):
        assert QSV_OP_DTYPE.fields[field][1] == offset
        assert getattr(_lib.QsvOp, field).offset == offset
    assert C.sizeof(_lib.QsvPlanConfig) == 20
    assert C.sizeof(_lib.QsvProfile) == 5 * 8 + 4 * 8 + 8 + 6 * 24  # six per-kernel arrays of three


def test_argument_errors_without_a_device():
    lib = _lib.load()
    n_words = C.c_size_t(0)
    ops = CircuitIR(3).cu3(0.1, 0.2, 0.3, 0, 1).packed().copy()
    assert lib.qsv_plan_build(3, 0, len(ops), _lib.as_ptr(ops), None, None, 0, C.byref(n_words)) == 0 and n_words.value > 0
    ops["target"][0] = 7  # out of range
    assert lib.qsv_plan_build(3, 0, len(ops), _lib.as_ptr(ops), None, None, 0, C.byref(n_words)) == _lib.QSV_E_ARG
    assert b"target" in lib.qsv_last_error(None)
    handle = C.c_void_p()
    assert lib.qsv_create(0, 0, 0, None, C.byref(handle)) == _lib.QSV_E_ARG
    assert lib.qsv_create(4, 9, 0, None, C.byref(handle)) == _lib.QSV_E_ARG


def test_no_cpu_fallback():
    """Without a GPU the evaluator must raise; on a GPU box this test is skipped."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from queasars_amd.circuit_evaluation import CircuitEvaluatorException, OperatorCircuitEvaluator

    with pytest.raises(CircuitEvaluatorException, match="no HIP device|hip"):
        OperatorCircuitEvaluator(PauliOperator(["ZZ"], [1.0]))
