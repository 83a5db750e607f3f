"""One population through the estimator evaluator, repeated (for rocprofv3 --kernel-trace): usage evalprof.py n P L"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L = (int(v) for v in sys.argv[1:4])
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=3))
for _ in range(20):
    ev.evaluate_circuits(circuits, params)
