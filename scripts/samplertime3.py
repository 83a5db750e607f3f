"""Sampler branch on whole populations: evaluations/s of the device-CVaR call next to the estimator call."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator, OperatorCircuitEvaluator

for n, P, L in ((12, 64, 2), (16, 64, 4), (20, 64, 4), (24, 32, 4)):
    _, circuits, params = helpers.population_circuits(n, L, P, seed=0)
    op = helpers.random_ising_operator(n, seed=3)
    ev = OperatorSamplerCircuitEvaluator(1024, op, alpha=0.5, seed=1)
    ex = OperatorCircuitEvaluator(op)
    for _ in range(3):
        ev.evaluate_circuits(circuits, params); ex.evaluate_circuits(circuits, params)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps): ev.evaluate_circuits(circuits, params)
    t_s = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): ex.evaluate_circuits(circuits, params)
    t_e = (time.perf_counter() - t0) / reps
    print(f"n={n} P={P} L={L}: sampler {t_s*1e6:.0f} us/call = {P/t_s:.0f} evals/s; estimator {t_e*1e6:.0f} us/call = {P/t_e:.0f} evals/s", flush=True)
