"""The headline's weak-scaling blocks on ONE GPU: bench.py --gpus N evaluates the population of 64 N individuals (seed 0), rank r its
block [64 r, 64 r + 64).  Here every block of the N = 8 population is evaluated alone, one after the other: routes, cost-model
imbalance, microseconds per step -- the slowest block is what a step of eight ranks waits for.
    python scripts/weak_blocks.py [N]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation
from queasars_amd.distributed import partition_by_cost, contiguous_shares, imbalance

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, per, L = 20, 64, 4
pop = EVQEPopulation.random_population(n, L, per * world, True, 0)
circuits = [ind.get_parameterized_quantum_circuit() for ind in pop.individuals]
params = [list(ind.parameter_values) for ind in pop.individuals]
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, 2020))
costs = ev.circuit_costs(circuits)
us = [c["microseconds"] for c in costs]
width = max(len(p) for p in params)
m = np.zeros((len(params), width))
for i, p in enumerate(params): m[i, : len(p)] = p
matrix = torch.from_numpy(m).cuda(); torch.cuda.synchronize()

def step_us(idx, reps=200):
    cs = [circuits[i] for i in idx]
    rows = matrix[idx[0]: idx[-1] + 1] if idx == list(range(idx[0], idx[-1] + 1)) else matrix.index_select(0, torch.as_tensor(idx, device="cuda")).contiguous()
    for _ in range(20): ev.evaluate_circuits(cs, rows)
    t0 = time.perf_counter()
    for _ in range(reps): ev.evaluate_circuits(cs, rows)
    return (time.perf_counter() - t0) / reps * 1e6

for name, shares in (("contiguous blocks", contiguous_shares(len(circuits), world)), ("dealt by cost", partition_by_cost(us, world))):
    print(f"== {name}: cost-model imbalance {imbalance(shares, us):.3f}")
    times = []
    for r, share in enumerate(shares):
        routes = {}
        for i in share:
            key = (costs[i]["route"], costs[i]["n_keys"])
            routes[key] = routes.get(key, 0) + 1
        t = step_us(list(share))
        times.append(t)
        print(f"  rank {r}: {len(share)} individuals, {t:6.1f} us per step, model {sum(us[i] for i in share):7.1f} us, routes {sorted(routes.items())}")
    print(f"  slowest {max(times):.1f} us, mean {np.mean(times):.1f} us -> a step of {world} ranks >= {max(times):.1f} us + exchange")
