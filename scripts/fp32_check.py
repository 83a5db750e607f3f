"""fp32 side tables against fp64 on split populations of several sizes (estimator, quadratic and general operators; sampler means)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from queasars_amd import workloads as helpers
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

for n in (14, 16, 20, 24, 26, 28):
    P = 16 if n < 26 else 6
    _, circuits, params = helpers.population_circuits(n, 4, P, seed=n)
    for name, op in (("ising", helpers.random_ising_operator(n, seed=1)), ("pauli", helpers.random_pauli_operator(n, 20, seed=2))):
        if name == "pauli" and n > 24:
            continue
        a = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
        b = np.asarray(OperatorCircuitEvaluator(op, dtype="fp32").evaluate_circuits(circuits, params))
        bound = 2e-6 * float(np.abs(op.coeffs).sum())
        print(f"n={n} {name}: max|fp32 - fp64| = {np.abs(a - b).max():.2e} (bound {bound:.2e})", flush=True)
        assert np.abs(a - b).max() < bound
print("ok")
