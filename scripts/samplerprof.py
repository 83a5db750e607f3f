"""One population through the sampler evaluator, repeated (for rocprofv3 --kernel-trace): usage samplerprof.py n P L shots"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator

n, P, L, shots = (int(v) for v in sys.argv[1:5])
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorSamplerCircuitEvaluator(shots, helpers.random_ising_operator(n, seed=3), alpha=0.5, seed=1)
for _ in range(20):
    ev.evaluate_circuits(circuits, params)
