"""The one-launch route with half sides, over and over: every repetition must give the bits of the first (a half side that read its
partner's rows before they had landed, or a stale line of them, would show as a repetition that differs).
    python scripts/soak_halves.py [repetitions]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = 20
op = helpers.random_ising_operator(n, seed=2020)
for layers in (4, 5):
    _, circuits, params = helpers.population_circuits(n, layers, 64, seed=0)
    ev = OperatorCircuitEvaluator(op)
    first = ev.evaluate_circuits(circuits, params)
    assert all(v == v for v in first)
    t0 = time.perf_counter()
    bad = 0
    for rep in range(reps):
        if ev.evaluate_circuits(circuits, params) != first:
            bad += 1
    print(f"L = {layers}: {reps} repetitions in {time.perf_counter() - t0:.1f} s, {bad} differ", flush=True)
    assert bad == 0
