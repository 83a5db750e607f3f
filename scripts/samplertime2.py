"""Where a sampler-branch call goes (n = 12 and 18, P = 20, 512 shots): device sampling vs host CVaR."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator, OperatorCircuitEvaluator
from queasars_amd.circuit_evaluation.circuit_evaluation import _cvar_of_samples

for n in (12, 18):
    P = 20
    _, circuits, params = helpers.population_circuits(n, 2, P, seed=0)
    op = helpers.random_ising_operator(n, seed=3)
    ev = OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=1)
    ex = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    for _ in range(3):
        ev.evaluate_circuits(circuits, params); ex.evaluate_circuits(circuits, params)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): ev.evaluate_circuits(circuits, params)
    t_all = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): states, values = dev.sample_batch(circuits, params, 512, 7, with_values=True)
    t_dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): [_cvar_of_samples(row, 0.5) for row in values]
    t_cvar = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): ex.evaluate_circuits(circuits, params)
    t_est = (time.perf_counter() - t0) / reps
    print(f"n={n}: sampler call {t_all*1e6:.0f} us = sample_batch {t_dev*1e6:.0f} + host CVaR {t_cvar*1e6:.0f} (+ glue); estimator call {t_est*1e6:.0f} us")
