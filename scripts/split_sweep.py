"""Randomised agreement check of the split path against the multi-pass path (GPU): populations of many sizes and depths."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads as helpers
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, StatevectorDevice


def device(n, split, dtype="fp64"):
    os.environ["QSV_SPLIT"] = "1" if split else "0"
    try:
        return StatevectorDevice(n, dtype=dtype)
    finally:
        del os.environ["QSV_SPLIT"]


worst = 0.0
for n in [int(a) for a in sys.argv[1:]] or list(range(13, 27)):
    op = helpers.random_ising_operator(n, seed=100 + n)
    dev_s, dev_p = device(n, True), device(n, False)
    ev_s, ev_p = OperatorCircuitEvaluator(op, statevector_device=dev_s), OperatorCircuitEvaluator(op, statevector_device=dev_p)
    for layers in (2, 3, 4, 5, 6):
        count = 12 if n <= 22 else 4
        _, circuits, params = helpers.population_circuits(n, layers, count, seed=1000 * n + layers)
        a = np.asarray(ev_s.evaluate_circuits(circuits, params))
        b = np.asarray(ev_p.evaluate_circuits(circuits, params))
        err = float(np.abs(a - b).max())
        worst = max(worst, err)
        print(f"n={n} L={layers} count={count} max|dE|={err:.2e}", flush=True)
        assert err < 1e-10, (n, layers, err)
    dev_s.close(); dev_p.close()
print("worst", worst)
