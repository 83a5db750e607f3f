"""Host-side stage times of one population evaluation (n=20, P=64): where the wall time outside the kernels goes."""
import sys, time
from itertools import chain
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd import _lib
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L = 20, 64, 4
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020))
dev = ev.statevector_device
lib, handle = dev._lib, dev._handle
for _ in range(3):
    ev.evaluate_circuits(circuits, params)
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
reps = 50
for _ in range(reps):
    t = time.perf_counter()
    ids = np.fromiter((dev.circuit_id(c) for c in circuits), dtype=np.int32, count=P)
    counts = np.fromiter((len(p) for p in params), dtype=np.int64, count=P)
    t = tick("ids+counts", t)
    for i, c in enumerate(circuits):
        if counts[i] < c.num_parameters:
            raise ValueError
    t = tick("validate", t)
    out = np.empty(P)
    lib.qsv_eval_begin(handle, P, _lib.as_ptr(ids), _lib.as_ptr(counts))
    t = tick("begin", t)
    values = np.fromiter(chain.from_iterable(params), dtype=np.float64, count=int(counts.sum()))
    t = tick("convert", t)
    lib.qsv_eval_push(handle, 0, P, _lib.as_ptr(values))
    t = tick("push", t)
    lib.qsv_eval_end(handle, _lib.as_ptr(out))
    t = tick("end(wait)", t)
    res = out.tolist()
    t = tick("tolist", t)
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:12s} {v / reps * 1e6:8.1f} us")
print(f"{'total':12s} {tot / reps * 1e6:8.1f} us")
