"""cProfile of the Python layer around one population evaluation (n=20, P=64): host time that is not GPU wait."""
import cProfile, pstats, sys, io
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

n, P, L = 20, 64, 4
_, circuits, params = helpers.population_circuits(n, L, P, seed=0)
ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020))
for _ in range(5):
    ev.evaluate_circuits(circuits, params)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    ev.evaluate_circuits(circuits, params)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(14)
print(out.getvalue())
