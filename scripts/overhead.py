"""Where a population evaluation's wall time goes: device time (events) vs host time, lists vs arrays."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L = 20, 64, 4
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020))
dev = ev.statevector_device
arr = [np.asarray(p) for p in params]
for label, pv in (("lists", params), ("arrays", arr)):
    ev.evaluate_circuits(circuits, pv)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): ev.evaluate_circuits(circuits, pv)
    wall = (time.perf_counter() - t0) / reps
    dev.set_profiling(True)
    ev.evaluate_circuits(circuits, pv)
    prof = dev.profile(); dev.set_profiling(False)
    print(f"{label}: wall {wall*1e3:.3f} ms  device total {prof['total_ms']:.3f} ms  passes {prof['pass_ms']:.3f} ms  expect {prof['expect_ms']:.3f} ms  launches {prof['n_pass_launches']}")
for g in (1, 2, 4):
    dev._push_groups = g
    t0 = time.perf_counter()
    for _ in range(20): ev.evaluate_circuits(circuits, params)
    print(f"push_groups={g}: wall {(time.perf_counter()-t0)/20*1e3:.3f} ms")
