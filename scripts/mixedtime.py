"""Populations that mix circuits with and without a split form (n = 20: five-layer individuals, and four-layer ones with a
few eight-layer ones among them): whole evaluator calls."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n = 20
op = helpers.random_ising_operator(n, seed=3)
ev = OperatorCircuitEvaluator(op)
_, five, p5 = helpers.population_circuits(n, 5, 64, seed=0)
_, four, p4 = helpers.population_circuits(n, 4, 58, seed=1)
_, eight, p8 = helpers.population_circuits(n, 8, 6, seed=2)
for name, cs, ps in (("64 five-layer circuits", five, p5), ("58 four-layer + 6 eight-layer circuits", four + eight, p4 + p8)):
    for _ in range(3):
        ev.evaluate_circuits(cs, ps)
    reps = 30
    t0 = time.perf_counter()
    for _ in range(reps):
        ev.evaluate_circuits(cs, ps)
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e6:.0f} us per call = {len(cs) / dt:.0f} evals/s", flush=True)
