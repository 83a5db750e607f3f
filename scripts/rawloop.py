"""The benchmark population through the C ABI with pre-packed arrays (no per-call Python work) next to the Python API:
what the host layer costs per step."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads, _lib
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L = 20, 64, 4
_, circuits, params = workloads.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
dev = ev.statevector_device
lib, h = dev._lib, dev._handle
ref = ev.evaluate_circuits(circuits, params)
ids = np.asarray([dev.circuit_id(c) for c in circuits], dtype=np.int32)
counts = np.asarray([len(p) for p in params], dtype=np.int64)
offsets = np.zeros(P + 1, dtype=np.int64); np.cumsum(counts, out=offsets[1:])
flat = np.concatenate([np.asarray(p) for p in params])
out = np.zeros(P)
reps = 200
def timed(fn):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: lib.qsv_eval_circuits(h, P, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(flat), _lib.as_ptr(out)))
assert out.tolist() == ref
print(f"qsv_eval_circuits, one push, pre-packed:        {t * 1e6:7.1f} us per population = {P / t:8.0f} evals/s")
half = P // 2
def two_halves():
    lib.qsv_eval_begin(h, P, _lib.as_ptr(ids), _lib.as_ptr(counts))
    lib.qsv_eval_push(h, 0, half, _lib.as_ptr(flat))
    lib.qsv_eval_push(h, half, P - half, C.c_void_p(flat.ctypes.data + 8 * int(offsets[half])))
    lib.qsv_eval_end(h, _lib.as_ptr(out))
import ctypes as C
t = timed(two_halves)
assert out.tolist() == ref
print(f"begin / two pushes / end, pre-packed:           {t * 1e6:7.1f} us per population = {P / t:8.0f} evals/s")
t = timed(lambda: dev.expectation_values(circuits, params))
print(f"StatevectorDevice.expectation_values:           {t * 1e6:7.1f} us per population = {P / t:8.0f} evals/s")
t = timed(lambda: ev.evaluate_circuits(circuits, params))
print(f"OperatorCircuitEvaluator.evaluate_circuits:     {t * 1e6:7.1f} us per population = {P / t:8.0f} evals/s")
