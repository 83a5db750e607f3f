"""Host-side time line of one population evaluation in two pushes (n = 20, P = 64): how long each C call takes."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd import _lib
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.circuit_evaluation.circuit_evaluation import _pack_doubles

n, P, L = 20, 64, 4
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020))
dev = ev.statevector_device
lib, handle = dev._lib, dev._handle
for _ in range(50):
    ev.evaluate_circuits(circuits, params)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
reps = 500
ids, need = dev._batch_metadata(circuits)
counts = np.fromiter(map(len, params), dtype=np.int64, count=P)
out = np.empty(P)
packed = [_pack_doubles(params[a:b], int(counts[a:b].sum())) for a, b in ((0, 32), (32, 64))]
t_all = time.perf_counter()
for _ in range(reps):
    t = time.perf_counter()
    lib.qsv_eval_begin(handle, P, _lib.as_ptr(ids), _lib.as_ptr(counts))
    t = tick("1 begin", t)
    for k, (a, b) in enumerate(((0, 32), (32, 64))):
        lib.qsv_eval_push(handle, a, b - a, _lib.as_ptr(packed[k]))
        t = tick(f"2 push {k}", t)
    lib.qsv_eval_end(handle, _lib.as_ptr(out))
    t = tick("3 end (wait)", t)
total = (time.perf_counter() - t_all) / reps
for k in sorted(acc):
    print(f"{k:16s} {acc[k] / reps * 1e6:8.1f} us")
print(f"total {total * 1e6:.1f} us per population (pre-packed parameters)")
t0 = time.perf_counter()
for _ in range(reps):
    ev.evaluate_circuits(circuits, params)
print(f"evaluate_circuits: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per population")
