"""The reference's calling pattern: population_size Python threads, one circuit per call -- plain, coalesced in Python,
coalesced inside the library (qsv_eval_coalesced)."""
import sys, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import CoalescingCircuitEvaluator, OperatorCircuitEvaluator

n, P, L = 20, 64, 4
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020))
ref = ev.evaluate_circuits(circuits, params)
cases = [("one call per thread, serialised", ev), ("coalesced in Python, 0.2 ms window", CoalescingCircuitEvaluator(ev, 2e-4, native=False)),
         ("coalesced in the library", CoalescingCircuitEvaluator(ev))]
for name, e in cases:
    with ThreadPoolExecutor(max_workers=P) as pool:
        for _ in range(3):
            list(pool.map(lambda j: e.evaluate_circuits([circuits[j]], [params[j]]), range(P)))
        t0 = time.perf_counter(); reps = 30
        for _ in range(reps):
            got = list(pool.map(lambda j: e.evaluate_circuits([circuits[j]], [params[j]])[0], range(P)))
        dt = (time.perf_counter() - t0) / reps
    assert got == ref
    print(f"{name}: {dt * 1e3:.2f} ms per population = {P / dt:.0f} evals/s")
