import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
os.environ["QSV_SPLIT_DEBUG"] = "1"
from queasars_amd import workloads as w
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
n, L, P = (int(x) for x in sys.argv[1:4])
_, circuits, params = w.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(w.random_ising_operator(n, seed=2020))
ev.evaluate_circuits(circuits, params)
