"""Throughput of whole evaluator calls by register size: estimator (Ising operator) and sampler branch (1024 shots, CVaR 0.5),
populations of 64 four-layer individuals (32 from 26 qubits on).  usage: nsweep.py [first last]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, OperatorSamplerCircuitEvaluator

first, last = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (8, 28)
for n in range(first, last + 1, 2):
    P = 64 if n < 26 else 32
    _, circuits, params = helpers.population_circuits(n, 4, P, seed=0)
    op = helpers.random_ising_operator(n, seed=3)
    ex = OperatorCircuitEvaluator(op)
    sa = OperatorSamplerCircuitEvaluator(1024, op, alpha=0.5, seed=1, statevector_device=ex.statevector_device)
    line = f"n={n:2d} P={P}:"
    for name, ev in (("estimator", ex), ("sampler", sa)):
        for _ in range(3):
            ev.evaluate_circuits(circuits, params)
        reps = 30 if n < 26 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            ev.evaluate_circuits(circuits, params)
        dt = (time.perf_counter() - t0) / reps
        line += f"  {name} {dt * 1e6:8.0f} us/call = {P / dt:9.0f} evals/s"
    print(line, flush=True)
    ex.statevector_device.close()
