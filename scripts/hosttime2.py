"""Host-side time line of one population evaluation with the production push plan (n = 20, P = 64): what happens
before the first launch and after the last kernel, i.e. the part of a step the GPU cannot hide."""
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
for _ in range(5):
    ev.evaluate_circuits(circuits, params)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
reps = 200
t_all = time.perf_counter()
for _ in range(reps):
    t = time.perf_counter()
    pairs = [(c, p) for c, p in zip(circuits, params) if c is not None and p is not None]
    cs, ps = [c for c, _ in pairs], [p for _, p in pairs]
    t = tick("1 evaluator prologue", t)
    ids, need = dev._batch_metadata(cs)
    counts = np.fromiter(map(len, ps), dtype=np.int64, count=P)
    bad = (counts < need).any()
    out = np.empty(P)
    t = tick("2 metadata+counts", t)
    lib.qsv_eval_begin(handle, P, _lib.as_ptr(ids), _lib.as_ptr(counts))
    t = tick("3 begin", t)
    head = P // 8
    bounds = [0, head, head + (P - head + 1) // 2, P]
    for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        chunk = ps[a:b]
        values = _pack_doubles(chunk, int(counts[a:b].sum()))
        t = tick(f"4 pack {k}", t)
        lib.qsv_eval_push(handle, a, b - a, _lib.as_ptr(values))
        t = tick(f"5 push {k}", t)
    lib.qsv_eval_end(handle, _lib.as_ptr(out))
    t = tick("6 end (wait)", t)
    res = [float(v) for v in out]
    t = tick("7 to list", t)
total = (time.perf_counter() - t_all) / reps
for k in sorted(acc):
    print(f"{k:24s} {acc[k] / reps * 1e6:8.1f} us")
print(f"{'total':24s} {total * 1e6:8.1f} us  -> {P / total:.0f} evals/s")
