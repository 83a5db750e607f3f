"""Stress: many Python threads evaluate random subsets of a mixed population (split and unsplittable circuits) on one
evaluator -- plain calls, coalesced calls, sampler calls -- and every value must equal the single-threaded result."""
import random, sys, time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from queasars_amd import workloads as helpers
from queasars_amd.circuit_evaluation import CoalescingCircuitEvaluator, OperatorCircuitEvaluator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
_, shallow, ps = helpers.population_circuits(n, 4, 24, seed=1)
_, deep, pd = helpers.population_circuits(n, 8, 8, seed=2)
circuits, params = shallow + deep, ps + pd
op = helpers.random_ising_operator(n, seed=3)
ev = OperatorCircuitEvaluator(op)
ref = ev.evaluate_circuits(circuits, params)
merged = CoalescingCircuitEvaluator(ev)
stop = time.time() + seconds
counts = [0] * 16


def worker(w):
    rng = random.Random(w)
    while time.time() < stop:
        if w % 3 == 0:
            j = rng.randrange(len(circuits))
            assert merged.evaluate_circuits([circuits[j]], [params[j]])[0] == ref[j]
            counts[w] += 1
        else:
            idx = [rng.randrange(len(circuits)) for _ in range(rng.randrange(1, 40))]
            got = ev.evaluate_circuits([circuits[j] for j in idx], [params[j] for j in idx])
            assert got == [ref[j] for j in idx]
            counts[w] += len(idx)
    return True


with ThreadPoolExecutor(max_workers=16) as pool:
    assert all(pool.map(worker, range(16)))
print(f"n={n}: {sum(counts)} evaluations from 16 threads in {seconds:.0f} s, all bitwise equal to the single-threaded values")
