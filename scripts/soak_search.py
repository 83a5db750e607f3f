"""Soak of the device-resident parameter search: 300 searches of 32 fresh structures each (shared circuits, embedded runs, the
notebooks' termination rule); resident set size and device memory every 100.  python scripts/soak_search.py"""
import os, sys, time, gc
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, psutil, torch
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation
from queasars_amd.evqe import solver as S
proc = psutil.Process(os.getpid())
def state(tag):
    free, total = torch.cuda.mem_get_info()
    print(f"{tag:40s} RSS {proc.memory_info().rss/2**20:8.1f} MiB  device in use {(total-free)/2**20:8.1f} MiB", flush=True)
n = 16
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
cfg = S.SPSA(termination_checker=S.SPSATerminationChecker(0.01, 2))
state("start")
for g in range(300):
    pop = EVQEPopulation.random_population(n, 3 + g % 3, 32, True, 100 + g)
    jobs = []
    for k, ind in enumerate(pop.individuals):
        run = cfg.new_run(ind.get_layer_parameter_values(-1), seed=k)
        run.embed = (np.asarray(ind.parameter_values, dtype=np.float64), np.asarray(ind.layer_parameter_indices[len(ind.layers)-1], dtype=np.int64))
        jobs.append((ind.get_parameterized_quantum_circuit(shared=True), run))
    S._minimize_batched(ev, jobs, on_device=True)
    if g % 100 == 99:
        gc.collect(); state(f"after {g+1} device-resident searches")
