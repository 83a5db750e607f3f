"""Whole-call time of one population by number of pushes (QSV_PUSHES in the environment): usage pushsweep.py n P L"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L = (int(v) for v in sys.argv[1:4])
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=3))
t_end = time.perf_counter() + 0.15
while time.perf_counter() < t_end:
    ev.evaluate_circuits(circuits, params)
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    ev.evaluate_circuits(circuits, params)
dt = (time.perf_counter() - t0) / reps
print(f"n={n} P={P} pushes={os.environ.get('QSV_PUSHES', 'suggested')}: {dt * 1e6:.0f} us per call = {P / dt:.0f} evals/s", flush=True)
