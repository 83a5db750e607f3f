"""Time of the sampler-branch evaluator (device sampling + CVaR) per call, and its kernel timeline under rocprofv3."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator, OperatorCircuitEvaluator

n, P = 12, 20
_, circuits, params = helpers.population_circuits(n, 2, P, seed=0)
op = helpers.random_ising_operator(n, seed=3)
ev = OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=1)
ex = OperatorCircuitEvaluator(op)
for _ in range(3):
    ev.evaluate_circuits(circuits, params); ex.evaluate_circuits(circuits, params)
for name, e in (("sampler", ev), ("estimator", ex)):
    t0 = time.perf_counter()
    for _ in range(50):
        e.evaluate_circuits(circuits, params)
    dt = (time.perf_counter() - t0) / 50
    print(f"{name}: {dt * 1e6:.0f} us per call of {P} circuits = {P / dt:.0f} evals/s")
