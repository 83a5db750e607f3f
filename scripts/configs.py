"""Runs the BASELINE.json configurations that fit one GPU and prints one JSON line per configuration."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

from queasars_amd import workloads as helpers  # noqa: E402
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator  # noqa: E402


def timed(evaluator, circuits, params, reps):
    evaluator.evaluate_circuits(circuits, params)
    t0 = time.perf_counter()
    for _ in range(reps):
        values = evaluator.evaluate_circuits(circuits, params)
    return (time.perf_counter() - t0) / reps, values


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,5")
    ap.add_argument("--terms5", type=int, default=500)
    args = ap.parse_args()
    todo = {int(c) for c in args.configs.split(",")}
    if 1 in todo:
        _, circuits, params = helpers.population_circuits(8, 2, 4, seed=0)
        op = helpers.random_pauli_operator(8, 20, seed=1234)
        ev = OperatorCircuitEvaluator(op)
        dt, values = timed(ev, circuits, params, 50)
        print(json.dumps({"config": 1, "n": 8, "P": 4, "terms": 20, "evals_per_s": 4 / dt, "us_per_call": dt * 1e6}), flush=True)
    if 2 in todo:
        _, circuits, params = helpers.population_circuits(20, 4, 64, seed=0)
        op = helpers.random_ising_operator(20, seed=2020)
        ev = OperatorCircuitEvaluator(op)
        dt, values = timed(ev, circuits, params, 10)
        print(json.dumps({"config": 2, "n": 20, "P": 64, "terms": len(op), "evals_per_s": 64 / dt, "ms_per_population": dt * 1e3}), flush=True)
    if 3 in todo:
        # one GPU's share of the 256-individual population: 32 individuals
        _, circuits, params = helpers.population_circuits(24, 4, 256, seed=0)
        circuits, params = circuits[:32], params[:32]
        op = helpers.random_ising_operator(24, seed=2024)
        ev = OperatorCircuitEvaluator(op)
        dt, values = timed(ev, circuits, params, 3)
        dev = ev.statevector_device
        dev.set_profiling(True)
        ev.evaluate_circuits(circuits, params)
        prof = dev.profile()
        kernels = [{"launches": prof["kernel_launches"][k], "avg_launch_us": 1e3 * prof["kernel_ms"][k] / max(1, prof["kernel_launches"][k]),
                    "algorithmic_GBps": prof["kernel_bytes"][k] / max(prof["kernel_ms"][k], 1e-9) / 1e6,
                    "moved_state_GBps": prof["kernel_moved_bytes"][k] / max(prof["kernel_ms"][k], 1e-9) / 1e6} for k in (0, 1)]
        print(json.dumps({"config": 3, "n": 24, "P_per_gpu": 32, "terms": len(op), "evals_per_s": 32 / dt, "ms_per_population": dt * 1e3,
                          "pass_kernels": kernels,
                          "note": "per instantiation (pass 0 / later passes): HIP events around every launch; launches of the "
                          "two streams overlap, so their times add up to more than the wall clock"}), flush=True)
    if 5 in todo:
        n = 28
        _, circuits, params = helpers.population_circuits(n, 4, 1, seed=0)
        op = helpers.random_pauli_operator(n, args.terms5, seed=2028)
        out = {"config": 5, "n": n, "terms": len(op)}
        values = {}
        for dtype in ("fp64", "fp32"):
            ev = OperatorCircuitEvaluator(op, dtype=dtype)
            dt, v = timed(ev, circuits, params, 1)
            values[dtype] = v[0]
            out[f"{dtype}_s_per_eval"] = dt
            del ev
        out["fp64_value"] = values["fp64"]
        out["abs_diff_fp32_vs_fp64"] = abs(values["fp32"] - values["fp64"])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
