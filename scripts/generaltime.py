"""General operators (config 5's family: random Pauli strings over {I, X, Y, Z}) on a small population: the term kernel on
split circuits (QSV_FACTOR=1, default) against the state + grouped-expectation path (QSV_FACTOR=0).
usage: generaltime.py n P L terms"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L, T = (int(v) for v in sys.argv[1:5])
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
op = helpers.random_pauli_operator(n, T, seed=2028)
t0 = time.perf_counter()
ev = OperatorCircuitEvaluator(op, dtype=sys.argv[5] if len(sys.argv) > 5 else "fp64")
print(f"set-up {time.perf_counter() - t0:.3f} s", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    vals = ev.evaluate_circuits(circuits, params)
    print(f"n={n} P={P} T={T}: call {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
