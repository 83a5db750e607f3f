import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
import helpers
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
t0 = time.perf_counter()
def lap(msg):
    global t0
    print(f"{msg}: {time.perf_counter() - t0:.2f} s", flush=True); t0 = time.perf_counter()
c_oracle = helpers.load_c_oracle(); lap("load oracle")
n = 20
_, circuits, params = helpers.population_circuits(n, 4, 2, seed=0); lap("population")
op = helpers.random_pauli_operator(n, 500, seed=2028); lap("operator")
ref = np.asarray([c_oracle.evaluate(c, p, op) for c, p in zip(circuits, params)]); lap("oracle evaluate")
ev = OperatorCircuitEvaluator(op); lap("evaluator")
got64 = np.asarray(ev.evaluate_circuits(circuits, params)); lap("evaluate fp64")
print(np.abs(got64 - ref).max())
ev32 = OperatorCircuitEvaluator(op, dtype="fp32"); lap("evaluator fp32")
got32 = np.asarray(ev32.evaluate_circuits(circuits, params)); lap("evaluate fp32")
