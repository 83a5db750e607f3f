#!/usr/bin/env python3
"""Headline benchmark: circuit-evals/sec of an EVQE population on MI355X.

Workload of the headline number (BASELINE.json configs[1], SURVEY.md 8(d) "Config 2"): n = 20 qubits, P = 64
individuals per GPU, L = 4 layers, genomes from the restated ``EVQEPopulation.random_population(..., random_seed=0)``,
random Ising Hamiltonian (190 ZZ + 20 Z terms, J, h ~ N(0,1), default_rng(2020)), fp64.  One "step" = one fitness
evaluation of the whole population through the product's ``evaluate_population_sharded`` (each rank evaluates its
block and every rank ends up with all fitness values: on one node through a table in shared host memory that the kernels
store into, no collective in the step; the RCCL all-gather is the way across nodes, the fallback, and what the first step is
checked against).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Weak scaling: the population has 64 * N individuals, rank r owns [64r, 64r+64); value = 64 * N * K / T.
Rank 0 prints ONE JSON line.  Beside the contract's fields it carries

  roofline          per instantiation of the gate-pass kernel (HIP events around every launch, measured in extra
                    profiled steps AFTER the timed region), and ``microbench``: single-gate sweeps at n = 24 and 26
  config3           BASELINE.json configs[2] at this N: n = 24, P = 256 in total (strong scaling), per-rank times
  cold_structure_evals_per_s / threaded_b1_evals_per_s   the reference's real calling patterns (N = 1 only)
  sampler_branch    the same population through the sampler evaluator (1024 shots, CVaR 0.5; N = 1 only)
  cpu_baseline      the plain-C oracle on the host cores, the NumPy oracle, and Qiskit Aer when importable (N = 1 only)
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_QUBITS = int(os.environ.get("QSV_BENCH_QUBITS", 20))
POP_PER_GPU = int(os.environ.get("QSV_BENCH_POP", 64))
N_LAYERS = int(os.environ.get("QSV_BENCH_LAYERS", 4))
PREWARM_S = float(os.environ.get("QSV_BENCH_PREWARM_S", 0.25))  # untimed load before the timed steps (see main)
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy rate)
FP64_PEAK_TFLOPS = 78.6     # vector fp64, half the 157.3 TFLOP/s fp32 vector rate of MI355X_MICROARCH.md (spec)


def kernel_names(n_qubits: int, factor: bool = True):
    """The kernels of the hot path at this size (register bits by size: qsv_api.hip resolve_config), in the order of
    qsv_profile's per-kernel arrays."""
    r = 4 if n_qubits >= 20 else 3
    return (f"qsv::pass_kernel<double, {r}, 2, true> (pass 0: synthesises a product state, writes only; for a split "
            "evaluation the two small virtual circuits, one workgroup each -- at 20 qubits and below in the one-launch "
            "instantiation pass_kernel<double, 3, 2, true, true>: sides of eight amplitudes per thread, a three-key side of "
            "thirteen qubits on two workgroups)",
            f"qsv::pass_kernel<double, {r}, 2, false> (later passes; the last one fuses the diagonal expectation and only reads)",
            "qsv::factor_moments_kernel<double> + qsv::factor_combine_kernel, timed as one (split evaluations under a "
            "quadratic diagonal operator: weighted Gram matrices of the two side tables, combined per evaluation; nothing of "
            "size 2^n is read)" if (factor and os.environ.get("QSV_FACTOR", "1") != "0") else
            "qsv::contract_kernel<double> (split evaluations: forms psi[i] from the two side tables on the fly and reduces "
            "sum_i D[i] |psi[i]|^2; reads D once per state)")


def ising_operator(n_qubits: int, seed: int):
    from queasars_amd.workloads import random_ising_operator

    return random_ising_operator(n_qubits, seed)


# ---- CPU baselines (rank 0, N = 1 only; the oracle is the checker and the baseline, never the product) ---------------


def cpu_baseline(circuits, params, operator, budget_s: float = 12.0, gpu_values=None):
    """Time the plain-C oracle (OpenMP, the host cores of this job) on a bounded sample of the same workload; also the
    NumPy oracle on two individuals, and Qiskit Aer's statevector estimator when it can be imported."""
    import helpers  # tests/helpers.py: the only place outside tests/ that touches oracle/, as the timed baseline

    orc = helpers.load_c_oracle()
    # the GPU box hands one job a 16-core share of the host: do not oversubscribe it
    orc.lib.qsvo_set_threads(helpers.host_cpu_share(16))
    cores = int(orc.lib.qsvo_max_threads())
    table = orc.diagonal_table(operator)  # once per operator, like qsv_set_operator on the GPU side: not timed
    scratch = np.zeros(2 << circuits[0].n_qubits, dtype=np.float64)
    orc.evaluate(circuits[0], params[0], operator, table, scratch)  # warm up (page faults, thread pool)
    done, t0 = 0, time.perf_counter()
    values = []
    while done < len(circuits) and (done < 2 or time.perf_counter() - t0 < budget_s):
        values.append(orc.evaluate(circuits[done], params[done], operator, table, scratch))
        done += 1
    elapsed = time.perf_counter() - t0
    out = {
        "value": done / elapsed,
        "unit": "circuit-evals/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {done} of the {len(circuits)} individuals of the same workload, plain-C oracle with OpenMP "
        f"({cores} threads), one sweep per gate, diagonal table prebuilt",
    }
    # NumPy oracle (single process): SURVEY.md 8(d) "CPU baseline" item 2
    t0 = time.perf_counter()
    numpy_values = [helpers.oracle_expectation(c, p, operator) for c, p in zip(circuits[:2], params[:2])]
    out["numpy_oracle"] = {"value": 2 / (time.perf_counter() - t0), "unit": "circuit-evals/s", "cores": 1,
                           "sample": "first 2 individuals, oracle/statevector_oracle.py (tensor-reshape formulation)",
                           "max_abs_diff_vs_c_oracle": float(np.abs(np.asarray(numpy_values) - np.asarray(values[:2])).max())}
    out["aer"] = aer_baseline(circuits, params, operator, gpu_values)
    return out, values


def aer_baseline(circuits, params, operator, gpu_values=None):
    """SURVEY.md 8(d) "CPU baseline" item 1: Qiskit Aer's statevector estimator on the same pubs, if the host has it.
    Where it runs it also pins parity: ``max_abs_diff_vs_aer`` = the GPU's values of the same individuals against Aer's
    (the reference's own arithmetic, circuit_evaluation.py:210-215, at precision 0)."""
    try:
        import qiskit_aer  # noqa: F401
        from qiskit import QuantumCircuit
        from qiskit.circuit.library import CU3Gate
        from qiskit.quantum_info import SparsePauliOp
        from qiskit_aer.primitives import EstimatorV2
    except Exception as exc:  # not installable offline
        return {"status": "unavailable", "reason": f"{type(exc).__name__}: {exc}"}
    pubs = []
    for c, p in list(zip(circuits, params))[:4]:
        qc = QuantumCircuit(c.n_qubits)
        for kind, target, control, theta, phi, lam in c.bound_ops(p):
            if kind == 1:
                qc.u(theta, phi, lam, target)
            elif kind == 2:
                qc.append(CU3Gate(theta, phi, lam), [control, target])
        pubs.append((qc, SparsePauliOp(operator.labels, operator.coeffs)))
    est = EstimatorV2(options={"backend_options": {"method": "statevector"}})
    est.run(pubs[:1], precision=0).result()
    t0 = time.perf_counter()
    result = est.run(pubs, precision=0).result()
    dt = time.perf_counter() - t0
    aer_values = [float(np.real(r.data.evs)) for r in result]
    out = {"status": "measured", "value": len(pubs) / dt, "unit": "circuit-evals/s", "cores": os.cpu_count(),
           "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "values": aer_values}
    if gpu_values is not None:
        out["max_abs_diff_vs_aer"] = float(np.abs(np.asarray(gpu_values[: len(aer_values)]) - np.asarray(aer_values)).max())
    return out


# ---- extra measurements around the headline ---------------------------------------------------------------------------


def kernel_rooflines(device, step, profiled_steps: int, traffic: dict, n_qubits: int = None):
    """Per kernel of the hot path (the two instantiations of the gate-pass kernel, the contraction kernel): launches, mean launch time (HIP events on the stream each launch runs
    on), algorithmic bytes and flops per launch, and the fractions of the two roofs.  Measured in steps of their own,
    after the timed region: the per-launch events cost a few microseconds each."""
    n_qubits = N_QUBITS if n_qubits is None else n_qubits
    device.set_profiling(True)
    acc = None
    for _ in range(profiled_steps):
        step()
        p = device.profile()
        if acc is None:
            acc = {k: (list(v) if isinstance(v, list) else v) for k, v in p.items()}
        else:
            for k, v in p.items():
                acc[k] = [a + b for a, b in zip(acc[k], v)] if isinstance(v, list) else acc[k] + v
    device.set_profiling(False)
    kernels = []
    for kind in (0, 1, 2):
        launches = acc["kernel_launches"][kind]
        if not launches or acc["kernel_ms"][kind] <= 0:
            continue
        avg_ms = acc["kernel_ms"][kind] / launches
        alg = acc["kernel_bytes"][kind] / launches
        moved = acc["kernel_moved_bytes"][kind] / launches
        if kind == 0:
            # a compact first pass writes 2^(m + k) amplitudes per state, not 2^n (plan.hpp COMPACT): the row is priced by what
            # the pass has to move in THIS design -- pricing 16 * 2^n per state gave fractions above 1
            alg = min(alg, moved)
        flops = acc["kernel_flops"][kind] / launches
        measured = traffic.get("kernels", {}).get(str(kind), {}).get("hbm_bytes_per_launch")
        entry = {
            "kernel": kernel_names(n_qubits)[kind],
            "launches": launches,
            "states_per_launch": acc["kernel_states"][kind] / launches,
            "avg_launch_us": avg_ms * 1e3,
            "algorithmic_bytes_per_launch": alg,
            "achieved_GBps": alg / (avg_ms * 1e-3) / 1e9,
            "frac_hbm_algorithmic": alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "moved_state_bytes_per_launch": moved,
            "traffic_bytes_per_launch": measured,
            "frac_hbm_measured_traffic": (measured / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if measured else None,
            "flops_per_launch": flops,
            "achieved_fp64_TFLOPs": flops / (avg_ms * 1e-3) / 1e12,
            "frac_fp64": flops / (avg_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
        }
        # what the counters say limits the kernel: neither roof is near -> a chain of dependent latencies
        entry["bound"] = "hbm" if entry["frac_hbm_algorithmic"] > 0.5 and (measured or 0) > 0.5 * alg else (
            "fp64" if entry["frac_fp64"] > 0.5 else "latency/issue")
        kernels.append(entry)
    return kernels, acc


def microbench_block():
    """SURVEY.md 8(d) config 3's microbenchmark, timed inside this run: one u and one cu3 per sweep of a 2^n fp64
    state (32 * 2^n algorithmic bytes per sweep), a few targets each, at n = 24 (256 MiB: the size of the Infinity
    Cache, so back-to-back sweeps are served on-die) and n = 26 (1 GiB: HBM)."""
    from queasars_amd.circuit_evaluation import StatevectorDevice

    out = {}
    for n in (24, 26):
        dev = StatevectorDevice(n, group=1)
        sweep_bytes = 32.0 * (1 << n)
        rows = []
        for target in sorted({0, 3, 7, 11, 12, 17, n - 3, n - 1}):
            control = (target + n // 2) % n
            ms_u = dev.bench_gate(target, -1, reps=20)
            ms_c = dev.bench_gate(target, control, reps=20)
            rows.append((target, sweep_bytes / ms_u / 1e6, sweep_bytes / ms_c / 1e6))
        rates = [r for _, u, c in rows for r in (u, c)]
        out[f"n{n}"] = {
            "state_MiB": (16 << n) >> 20,
            "targets": [t for t, _, _ in rows],
            "u_GBps": [round(u, 1) for _, u, _ in rows],
            "cu3_GBps": [round(c, 1) for _, _, c in rows],
            "min_GBps": min(rates), "mean_GBps": sum(rates) / len(rates), "max_GBps": max(rates),
            "frac_of_8TBps_mean": sum(rates) / len(rates) / HBM_PEAK_GBPS,
            "bound": "hbm" if n >= 26 else "infinity cache (state = 256 MiB)",
        }
        dev.close()
    out["note"] = ("achieved = 32 * 2^n bytes per sweep / mean sweep time (HIP events, 20 sweeps back to back); "
                   "the HBM-honest row is n = 26; north_star's 60 % target is quoted on n = 24")
    return out


def cold_and_threaded(operator, n_steps: int = 7):
    """The reference's calling patterns (N = 1): (a) every step brings a population of structures the device has never
    seen (selection after topological search / layer removal: plan building and upload are inside the timed region);
    (b) population_size threads, one circuit per call (selection.py:75-82), through CoalescingCircuitEvaluator."""
    from concurrent.futures import ThreadPoolExecutor

    from queasars_amd.circuit_evaluation import CoalescingCircuitEvaluator, OperatorCircuitEvaluator
    from queasars_amd.evqe import EVQEPopulation

    evaluator = OperatorCircuitEvaluator(operator)
    pops = []
    for s in range(n_steps + 1):
        pop = EVQEPopulation.random_population(N_QUBITS, N_LAYERS, POP_PER_GPU, True, 1000 + s)
        pops.append(([ind.get_parameterized_quantum_circuit() for ind in pop.individuals],
                     [list(ind.parameter_values) for ind in pop.individuals]))
    evaluator.evaluate_circuits(*pops[0])  # code objects, buffers, host worker threads
    step_times = []
    for circuits, params in pops[1:]:
        t0 = time.perf_counter()
        evaluator.evaluate_circuits(circuits, params)
        step_times.append(time.perf_counter() - t0)
    # median step: a single host hiccup (thread wake-up, page fault) in four 0.8 ms steps would otherwise dominate
    cold = POP_PER_GPU / sorted(step_times)[len(step_times) // 2]
    if os.environ.get("QSV_BENCH_VERBOSE"):
        print("cold step times (us):", [round(t * 1e6) for t in step_times], file=sys.stderr)
    circuits, params = pops[0]
    merged = CoalescingCircuitEvaluator(evaluator)
    with ThreadPoolExecutor(max_workers=POP_PER_GPU) as pool:
        def one(j):
            return merged.evaluate_circuits([circuits[j]], [params[j]])[0]

        for _ in range(3):
            list(pool.map(one, range(POP_PER_GPU)))
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            got = list(pool.map(one, range(POP_PER_GPU)))
        threaded = POP_PER_GPU * reps / (time.perf_counter() - t0)
        # what this calling pattern can reach on this host whatever the backend: the same 64-thread pool.map with tasks
        # that do nothing (CPython's executor and thread hand-over are the arrival rate of the real calls)
        t0 = time.perf_counter()
        for _ in range(reps):
            list(pool.map(lambda j: j, range(POP_PER_GPU)))
        noop = POP_PER_GPU * reps / (time.perf_counter() - t0)
        # ... and with tasks that BLOCK outside the interpreter for about as long as a merged evaluation takes (0.1 ms of
        # sleep with the GIL released): every task then costs thread hand-overs, as the real calls do
        t0 = time.perf_counter()
        for _ in range(reps):
            list(pool.map(lambda j: time.sleep(1e-4), range(POP_PER_GPU)))
        blocking = POP_PER_GPU * reps / (time.perf_counter() - t0)
    assert got == evaluator.evaluate_circuits(circuits, params)
    evaluator.statevector_device.close()
    return cold, threaded, noop, blocking


def sampler_block(operator, circuits, params, shots: int = 1024, alpha: float = 0.5, reps: int = 20):
    """The reference's sampler branch (OperatorSamplerCircuitEvaluator, circuit_evaluation.py:147-157) on the same
    population: `shots` samples per circuit drawn, valued and reduced to CVaR_alpha on the device."""
    from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator

    evaluator = OperatorSamplerCircuitEvaluator(shots, operator, alpha=alpha, seed=1)
    for _ in range(3):
        evaluator.evaluate_circuits(circuits, params)
    t0 = time.perf_counter()
    for _ in range(reps):
        evaluator.evaluate_circuits(circuits, params)
    rate = len(circuits) * reps / (time.perf_counter() - t0)
    evaluator.statevector_device.close()
    return {"value": rate, "unit": "circuit-evals/sec", "shots": shots, "alpha": alpha,
            "note": "same population and operator as the headline, through the sampler evaluator (device-side sampling, "
                    "operator values and CVaR); wall clock of whole calls"}


def search_block(operator, population, reps: int = 5):
    """What the evaluations are made FOR: one EVQE last-layer parameter search of the benchmark population (one SPSA run per
    individual, the notebooks' 33 iterations of two evaluations, all runs in lock-step; reference: mutation.py:28-89, one
    optimiser per individual on a worker thread) -- with the optimiser's state on the device (evqe/device_search.py,
    qsv_spsa_step) and with the whole-array driver on the host."""
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.evqe import solver as S

    evaluator = OperatorCircuitEvaluator(operator)
    cfg = S.SPSA()

    def jobs(shared: bool):
        # (as the driver builds them, solver._optimize_layers: on the device the individual's fully parameterised circuit --
        # one shared object per structure -- with the last layer's entries moving inside the full parameter vector; on the host
        # a fresh circuit per search with the other layers' values bound into it, as the reference has it)
        out = []
        for k, ind in enumerate(population.individuals):
            run = cfg.new_run(ind.get_layer_parameter_values(-1), seed=k)
            if shared:
                run.embed = (np.asarray(ind.parameter_values, dtype=np.float64),
                             np.asarray(ind.layer_parameter_indices[len(ind.layers) - 1], dtype=np.int64))
                out.append((ind.get_parameterized_quantum_circuit(shared=True), run))
            else:
                out.append((ind.get_partially_parameterized_quantum_circuit({-1}), run))
        return out

    out = {"individuals": len(population.individuals), "iterations": cfg.maxiter, "unit": "circuit-evals/s",
           "note": "wall clock of whole searches as the EVQE driver runs them (host: the 64 freshly bound circuits are registered "
                   "inside the search; device: the structures' shared circuits are registered once); best of five"}
    final = {}
    for name, on_device in (("state_on_device", True), ("whole_array_on_host", False)):
        S._minimize_batched(evaluator, jobs(on_device), on_device=on_device)
        best, evals = None, 0
        for _ in range(reps):
            j = jobs(on_device)
            t0 = time.perf_counter()
            S._minimize_batched(evaluator, j, on_device=on_device)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            evals = sum(run.nfev for _, run in j)
        final[name] = np.concatenate([run.x for _, run in j])
        out[name] = {"ms_per_search": best * 1e3, "value": evals / best, "evaluations": evals}
    out["max_abs_diff_of_final_iterates"] = float(np.abs(final["state_on_device"] - final["whole_array_on_host"]).max())
    evaluator.statevector_device.close()
    return out


def trajectory_block(operator, generations: int = 8, search_precision: str = None):
    """What a user sits in: the REAL solver on the benchmark's operator (n = 20, population 64, two initial layers, the notebook's
    optimiser and mutation probabilities, examples/evqe_jssp_optimization.ipynb: SPSA 33 iterations, parameter search 0.39,
    topological search 0.79, layer removal 0.02), `generations` generations.  The population's layers grow every generation
    (base/evolving_ansatz_minimum_eigensolver.py:401-433, mutation.py:347-353), so the evaluations per second drift from the
    shallow-circuit figure of the headline to the deep-circuit one.  Per generation: wall-clock evaluations per second (the
    last-layer search of everybody, the scoring, the previous generation's parameter searches), the layer histogram and the
    routes the scored population's circuits take on the device."""
    from collections import Counter

    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.evqe import solver as S

    evaluator = OperatorCircuitEvaluator(operator)
    # (search_precision = "fp32": the parameter searches on a single-precision handle of the same operator, the fitness as ever)
    searcher = OperatorCircuitEvaluator(operator, dtype=search_precision) if search_precision else None
    checker = S.SPSATerminationChecker(minimum_relative_change=0.01, allowed_consecutive_violations=2)
    cfg = S.EVQEMinimumEigensolverConfiguration(
        optimizer=S.SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True, termination_checker=checker),
        population_size=POP_PER_GPU, max_generations=generations, random_seed=0, n_initial_layers=2,
        randomize_initial_population_parameters=True, speciation_genetic_distance_threshold=1, use_tournament_selection=True,
        tournament_size=2, selection_alpha_penalty=0.15, selection_beta_penalty=0.02, parameter_search_probability=0.39,
        topological_search_probability=0.79, layer_removal_probability=0.02)
    rows, marks = [], {"t": time.perf_counter(), "evals": 0}

    def scored(info):
        now = time.perf_counter()
        individuals = info["population"].individuals
        circuits = [ind.get_parameterized_quantum_circuit(shared=True) for ind in individuals]
        routes = Counter(c["route"] for c in evaluator.circuit_costs(circuits))
        layers = Counter(len(ind.layers) for ind in individuals)
        dt, de = now - marks["t"], info["circuit_evaluations"] - marks["evals"]
        rows.append({"generation": info["generation"], "evaluations": de, "seconds": dt, "evals_per_s": de / dt,
                     "layers": {str(k): layers[k] for k in sorted(layers)}, "mean_layers": sum(k * v for k, v in layers.items()) / len(individuals),
                     "routes": dict(routes), "best_value": min(info["values"])})
        marks["t"], marks["evals"] = time.perf_counter(), info["circuit_evaluations"]  # (the bookkeeping above is not the solver's time)

    t0 = time.perf_counter()
    result = S.EVQEMinimumEigensolver(cfg, on_generation=scored).compute_minimum_eigenvalue(evaluator, searcher)
    total = time.perf_counter() - t0
    evaluator.statevector_device.close()
    if searcher is not None:
        searcher.statevector_device.close()
    return {"workload": f"EVQE on the {N_QUBITS}-qubit Ising operator of the headline, population {POP_PER_GPU}, from 2 layers, {generations} "
                        "generations; the notebook's optimiser and mutation probabilities",
            "generations": rows, "evaluations": sum(result.circuit_evaluations), "seconds": total,
            "evals_per_s": sum(result.circuit_evaluations) / total, "eigenvalue": result.eigenvalue,
            "kept_state_searches": os.environ.get("QSV_KEPT_STATES", "default (on where the circuit costs say it pays)"),
            "search_precision": search_precision or "fp64",
            "unit": "circuit-evals/s",
            "note": "wall clock of the whole solver (its own Python included), one process, one GPU; generation g's evaluations = "
                    "the parameter searches of generation g - 1's survivors (every layer of 39 % of them), the last-layer search of "
                    "everybody and the scoring"}


def layer_search_block(reps: int = 3):
    """The depth rows where a run lives after a few generations (n = 20, P = 64, seven and eight layers), as a LAST-LAYER search
    evaluates them (mutation.py:57-59: the other layers bound): whole circuits (rounds 1-3) against kept states for the
    individuals without a split form (round 4, solver._kept_state_circuits).  Whole evaluator calls, expectation included."""
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.evqe import EVQEPopulation

    out = {}
    operator = ising_operator(N_QUBITS, 2020)
    reference = {}
    for layers, precision in ((6, "fp64"), (7, "fp64"), (8, "fp64"), (7, "fp32"), (8, "fp32")):
        population = EVQEPopulation.random_population(N_QUBITS, layers, POP_PER_GPU, True, 0)
        evaluator = OperatorCircuitEvaluator(operator, dtype=precision)
        whole = [ind.get_partially_parameterized_quantum_circuit({layers - 1}) for ind in population.individuals]
        values = [list(ind.get_layer_parameter_values(layers - 1)) for ind in population.individuals]
        costs = evaluator.circuit_costs(whole)
        pairs = [ind.get_layer_search_circuits(layers - 1) for ind in population.individuals]
        deep = [i for i, c in enumerate(costs) if c["route"] == "gate passes"]
        states = evaluator.keep_states([pairs[i][0] for i in deep], [[] for _ in deep])
        mixed = list(whole)
        for i, state in zip(deep, states):
            mixed[i] = pairs[i][1].continue_from(state)

        def rate(circuits):
            for _ in range(3):
                got = evaluator.evaluate_circuits(circuits, values)
            t0 = time.perf_counter()
            evaluator.evaluate_circuits(circuits, values)
            per = time.perf_counter() - t0
            n = max(5, int(0.3 / max(per, 1e-6)))
            best = 0.0
            for _ in range(reps):
                t0 = time.perf_counter()
                for _ in range(n):
                    evaluator.evaluate_circuits(circuits, values)
                best = max(best, len(circuits) * n / (time.perf_counter() - t0))
            return best, np.asarray(got)

        whole_rate, whole_values = rate(whole)
        kept_rate, kept_values = rate(mixed)
        row = {"whole_circuits_evals_per_s": whole_rate, "kept_states_evals_per_s": kept_rate, "gain": kept_rate / whole_rate,
               "individuals_on_kept_states": len(deep), "max_abs_diff": float(np.abs(whole_values - kept_values).max()),
               "routes_whole": {r: sum(c["route"] == r for c in costs) for r in sorted({c["route"] for c in costs})}}
        if precision == "fp64":
            reference[layers] = whole_values
        else:  # (a search tolerates single precision; the fitness behind it stays fp64: the error against the fp64 values)
            row["max_abs_diff_vs_fp64"] = float(np.abs(kept_values - reference[layers]).max())
            row["sum_abs_coefficients"] = float(np.abs(operator.coeffs).sum())
        out[f"L{layers}" + ("" if precision == "fp64" else "_fp32")] = row
        del states, mixed
        evaluator.statevector_device.close()
    out["note"] = ("last-layer search points of the whole population per call; kept states: the individuals whose circuits take gate "
                   "passes over the 2^n state evaluate from the state in front of the last layer (qsv_prefix_create), the others as "
                   "before.  A kept-state evaluation moves 16 MiB read + 16 write + 16 read + the diagonal table: HBM bound; "
                   "the _fp32 rows: the same on a single-precision handle (half the bytes; the solver does not switch precision "
                   "by itself)")
    return out


def config5_sweep_block(sizes=(24, 26, 28)):
    """BASELINE.json configs[4] as a sweep: one genome (four layers, seed 0), 500 random Pauli strings (default_rng(2028)), single
    and double precision at n = 24, 26, 28: |fp32 - fp64| of the expectation value, evaluations per second on the library's
    default route (the genome has a split form: the term kernel on two small states) and THROUGH THE 2^n STATE (splitting off:
    gate passes over the state, then one read of it per group of strings), and the later-pass kernel's launch time and
    fraction of the HBM roof there (HIP events; 2 * sizeof(amplitude) * 2^n bytes per pass)."""
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.evqe import EVQEPopulation
    from queasars_amd.workloads import random_pauli_operator

    rows = {}
    for n in sizes:
        individual = EVQEPopulation.random_population(n, 4, 1, True, 0).individuals[0]
        circuits, params = [individual.get_parameterized_quantum_circuit()], [list(individual.parameter_values)]
        operator = random_pauli_operator(n, 500, seed=2028)
        row = {}
        for precision in ("fp64", "fp32"):
            entry = {}
            for route, split in (("default_route", 1), ("through_the_state", 0)):
                evaluator = OperatorCircuitEvaluator(operator, dtype=precision)
                device = evaluator.statevector_device
                if not split:
                    device.set_option("split", 0)
                value = evaluator.evaluate_circuits(circuits, params)[0]
                t0 = time.perf_counter()
                evaluator.evaluate_circuits(circuits, params)
                per = time.perf_counter() - t0
                reps = max(2, min(50, int(0.5 / max(per, 1e-6))))
                t0 = time.perf_counter()
                for _ in range(reps):
                    evaluator.evaluate_circuits(circuits, params)
                entry[route] = {"value": value, "evals_per_s": reps / (time.perf_counter() - t0)}
                if not split:
                    device.set_option("streams", 1)
                    kernels, _ = kernel_rooflines(device, lambda: evaluator.evaluate_circuits(circuits, params), 3, {}, n)
                    later = next((k for k in kernels if "later passes" in k["kernel"]), None)
                    if later is not None:
                        entry[route].update({"later_pass_launch_us": later["avg_launch_us"], "later_pass_GBps": later["achieved_GBps"],
                                             "later_pass_frac_hbm": later["frac_hbm_algorithmic"], "later_pass_launches": later["launches"]})
                device.close()
            row[precision] = entry
        row["abs_diff_fp32_vs_fp64"] = abs(row["fp32"]["default_route"]["value"] - row["fp64"]["default_route"]["value"])
        row["abs_diff_fp32_vs_fp64_through_the_state"] = abs(row["fp32"]["through_the_state"]["value"] - row["fp64"]["through_the_state"]["value"])
        row["abs_diff_of_the_two_routes_fp64"] = abs(row["fp64"]["default_route"]["value"] - row["fp64"]["through_the_state"]["value"])
        row["sum_abs_coefficients"] = float(np.abs(operator.coeffs).sum())
        rows[f"n{n}"] = row
    rows["note"] = ("fp32 tolerance of the GPU tests: 2e-6 * sum |c_k|.  later_pass_*: pass_kernel<real, R, X, false> on the genome's own "
                    "gate passes (not a single-gate sweep), one state per launch")
    return rows


# ---- the deep (unsplit) multi-pass path: the statevector sweep north_star names ----------------------------------------

DEEP_ROWS = {
    # name: (qubits, layers, individuals, splitting on?, precision)   -- reference behaviour: circuit_evaluation.py:200-215
    "deep_n20_L8": (20, 8, 64, True, "fp64"),
    "deep_n24_L4_nosplit": (24, 4, 32, False, "fp64"),
    "deep_n24_L8": (24, 8, 32, True, "fp64"),
    # the same rows in single precision (BASELINE configs[4] asks for an fp32 / fp64 sweep): the generated fp32 round loop
    "deep_n20_L8_fp32": (20, 8, 64, True, "fp32"),
    "deep_n24_L4_nosplit_fp32": (24, 4, 32, False, "fp32"),
    "deep_n24_L8_fp32": (24, 8, 32, True, "fp32"),
}


def load_traffic(row: str) -> dict:
    """HBM traffic per launch of a bench row from the committed profile (profiles/r04_traffic.json: separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --only <row>`); not measured inside this run."""
    for name in ("r04_traffic.json", "r03_traffic.json", "traffic.json"):
        f = ROOT / "profiles" / name
        if f.exists():
            try:
                data = json.loads(f.read_text())
            except Exception:
                continue
            if row in data:
                return data[row]
            if row == "headline" and "kernels" in data:
                return data
    return {}


def deep_block(local_rank: int, only: str = None, kernels_only: bool = False):
    """Populations whose circuits the splitter cannot take (or may not: splitting switched off): every evaluation runs the
    multi-pass plan over its 2^n state.  Per row: whole-call throughput with the library's defaults, then -- on one HIP
    stream, so that a launch has the chip to itself -- per instantiation of the gate-pass kernel launches, mean launch
    time (HIP events), SURVEY 8(d) bytes, what the launches really move, and the fractions of the HBM and fp64 roofs."""
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.evqe import EVQEPopulation

    rows = {}
    for name, (n, layers, pop, split, precision) in DEEP_ROWS.items():
        if only is not None and name != only:
            continue
        population = EVQEPopulation.random_population(n, layers, pop, True, 0)
        circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
        params = [list(ind.parameter_values) for ind in population.individuals]
        operator = ising_operator(n, 2020 if n == 20 else 2024)
        evaluator = OperatorCircuitEvaluator(operator, device=local_rank, dtype=precision)
        device = evaluator.statevector_device
        if not split:
            device.set_option("split", 0)

        def step():
            return evaluator.evaluate_circuits(circuits, params)

        if kernels_only:  # (profiling runs: rocprofv3 then sees the one-stream launches alone)
            device.set_option("streams", 1)
        t0 = time.perf_counter()
        for _ in range(3):
            values = step()
        per = (time.perf_counter() - t0) / 3
        for _ in range(int(0.15 / max(per, 1e-6))):  # clocks under load before anything is timed
            step()
        reps = max(3, int(0.25 / max(per, 1e-6)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        rate = pop * reps / (time.perf_counter() - t0)
        # how the population is evaluated: ordinary plans (passes) / split
        device.set_profiling(True)
        step()
        prof = device.profile()
        device.set_profiling(False)
        # the kernels alone: one stream
        device.set_option("streams", 1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        rate_one = pop * reps / (time.perf_counter() - t0)
        kernels, _ = kernel_rooflines(device, step, 5, load_traffic(name), n)
        if precision == "fp32":
            for k in kernels:  # (kernel_names() spells the fp64 instantiations; the single-precision peak is twice the fp64 one)
                k["kernel"] = k["kernel"].replace("pass_kernel<double, 4, 2,", "pass_kernel<float, 3, 0,").replace("<double>", "<float>")
                k["frac_fp32"] = k.pop("frac_fp64") / 2.0
                k.pop("achieved_fp64_TFLOPs", None)
        later = next((k for k in kernels if "later passes" in k["kernel"]), None)
        rows[name] = {
            "workload": f"{n}-qubit EVQE population={pop}, {layers} layers, Ising operator, {precision}"
                        + ("" if split else ", register splitting switched off"),
            "value": rate, "unit": "circuit-evals/s", "value_one_stream": rate_one,
            "pass_launches_per_call": prof["n_pass_launches"], "state_passes_per_call": prof["n_state_passes"],
            "split_evaluations": int(prof["kernel_states"][2]),
            "kernels_one_stream": kernels,
            "later_pass_frac_hbm": later["frac_hbm_algorithmic"] if later else None,
            "later_pass_frac_fp64": later.get("frac_fp64") if later else None,
            "all_values_finite": bool(np.isfinite(values).all()),
        }
        evaluator.statevector_device.close()
    rows["note"] = ("gate application over the 2^n state on real EVQE circuits (reference: circuit_evaluation.py:200-215).  "
                    "kernels_one_stream: HIP events around every launch with the handle on ONE stream (a launch then has the "
                    "chip to itself; with the default two streams launches overlap and stretch each other), bytes per SURVEY "
                    "8(d): 16 * 2^n per state and direction a pass has to move; traffic_bytes_per_launch from the committed "
                    "rocprofv3 PMC passes of `bench.py --only <row>` (profiles/r04_traffic.json), not measured in this run")
    return rows



def agreed_count(mine: int, world: int, comm_device: str, limit: int = 5000) -> int:
    """A loop count every rank uses: rank 0's (loops whose steps are collectives must not end at different iterations on
    different ranks -- a time-based exit per rank is a deadlock waiting for two clocks to disagree)."""
    mine = max(0, min(int(mine), limit))
    if world <= 1:
        return mine
    t = torch.tensor([mine], dtype=torch.int64, device=comm_device)
    dist.broadcast(t, src=0)
    return int(t.item())


# what the exchange of the fitness values adds to a rank's step on one node: through the node's shared table (no collective: the
# kernels store into the rank's slot, queasars_amd/distributed.py) 2 - 7 us over the evaluation alone on a group of ONE rank
# (profiles/r04_gatherstep.txt: 57.2 us against 54.9 at the end of the round); the RCCL all-gather into host-mapped memory it
# replaced there: 21 - 27.  The prediction takes 7.
COLLECTIVE_US = 7.0


def config3_block(world: int, rank: int, local_rank: int, steps: int = 8, layers: int = 4):
    """BASELINE.json configs[2] as north_star states it: n = 24, L = 4, P = 256 IN TOTAL (strong scaling: at N = 8 rank r
    takes [32r, 32r+32), at N = 1 the one GPU evaluates all 256), 300-term Ising operator of default_rng(2024), through
    the product's ``evaluate_population_sharded``.  ``layers`` = 8: the same with eight-layer individuals, whose
    evaluations are milliseconds of gate passes per rank (config3_deep)."""
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.distributed import evaluate_population_sharded, shard_bounds
    from queasars_amd.evqe import EVQEPopulation

    n, total = 24, 256
    population = EVQEPopulation.random_population(n, layers, total, True, 0)
    circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
    params = [list(ind.parameter_values) for ind in population.individuals]
    operator = ising_operator(n, 2024)
    evaluator = OperatorCircuitEvaluator(operator, device=local_rank)
    lo, hi = shard_bounds(total, world, rank)

    t_warm = time.perf_counter()  # warm-up: plans, buffers, and 0.1 s of load for the clocks (as before the headline)
    for _ in range(3):
        evaluate_population_sharded(evaluator, circuits, params)
    comm_device = "cpu" if (world > 1 and dist.get_backend() != "nccl") else "cuda"
    for _ in range(agreed_count(int(0.1 / max((time.perf_counter() - t_warm) / 3, 1e-6)), world, comm_device)):
        evaluate_population_sharded(evaluator, circuits, params)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        values = evaluate_population_sharded(evaluator, circuits, params)
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("QSV_BENCH_VERBOSE"):
        print("config 3 step times (us):", [round((b - a) * 1e6) for a, b in zip([t0] + marks[:-1], marks)], file=sys.stderr)
    # this rank's own block alone, after the timed region: what the imbalance is computed from
    own_c, own_p = circuits[lo:hi], params[lo:hi]
    t1 = time.perf_counter()
    for _ in range(steps):
        evaluator.evaluate_circuits(own_c, own_p)
    own_s = (time.perf_counter() - t1) / steps
    assert len(values) == total and all(np.isfinite(values))
    per_rank = [own_s]
    if world > 1:
        t = torch.tensor([elapsed, own_s], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        elapsed = max(float(g[0]) for g in gathered)
        per_rank = [float(g[1]) for g in gathered]
    # What strong scaling can give (N = 1 only): the first rank's block at 2, 4, 8 ranks timed on this GPU, plus the
    # collective's latency as measured on a one-rank RCCL group -- skew between ranks and the ring's hops are not in it.
    predicted = None
    if world == 1:
        from queasars_amd.distributed import contiguous_shares, evaluation_costs, imbalance, partition_by_cost

        predicted = {}
        t_all = elapsed / steps
        costs = evaluation_costs(evaluator, circuits)
        for g in (2, 4, 8):
            # every rank's share (dealt by circuit cost where the contiguous blocks are uneven, queasars_amd/distributed.py) timed
            # on this GPU: the step is as long as the SLOWEST share
            shares = partition_by_cost(costs, g) if costs is not None else contiguous_shares(total, g)
            times = []
            for share in shares:
                bc, bp = [circuits[i] for i in share], [params[i] for i in share]
                evaluator.evaluate_circuits(bc, bp)
                t1 = time.perf_counter()
                for _ in range(steps):
                    evaluator.evaluate_circuits(bc, bp)
                times.append((time.perf_counter() - t1) / steps)
            slowest = max(times)
            predicted[str(g)] = {"slowest_share_ms": slowest * 1e3, "mean_share_ms": sum(times) / len(times) * 1e3,
                                 "imbalance_measured_max_over_mean": slowest / (sum(times) / len(times)),
                                 "imbalance_by_cost_model_dealt": imbalance(shares, costs) if costs is not None else None,
                                 "imbalance_by_cost_model_contiguous_blocks": imbalance(contiguous_shares(total, g), costs) if costs is not None else None,
                                 "dealt_by_cost": shares != contiguous_shares(total, g),
                                 "speedup": t_all / (slowest + COLLECTIVE_US * 1e-6)}
        predicted["note"] = (f"speedup at N ranks = this GPU's time for all 256 / (its time for the SLOWEST of the N shares + "
                             f"{COLLECTIVE_US:.0f} us for the exchange through the node's shared table, profiles/r04_gatherstep.txt); a prediction from one GPU, not a "
                             "measurement: no multi-GPU node was available to the builder")
    evaluator.statevector_device.close()
    return {
        "workload": f"24-qubit EVQE population = 256 in total, {layers} layers, Ising 300 terms (default_rng(2024)), fp64 "
        "(BASELINE.json configs[2]); strong scaling: rank r evaluates its share, the fitness values exchanged through the node's shared table (RCCL all-gather as fallback)",
        "predicted_speedup": predicted,
        "value": total * steps / elapsed, "unit": "circuit-evals/s", "n_gpus": world, "steps": steps,
        "ms_per_step": elapsed / steps * 1e3, "scaling": "strong", "individuals_per_rank": hi - lo,
        "per_rank_ms_own_block": [round(x * 1e3, 3) for x in per_rank],
        "imbalance_max_over_mean": max(per_rank) / (sum(per_rank) / len(per_rank)),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (200 steps of 85 us: a timed region of 17 ms; 20 steps made it 1.7 ms, a tenth of which one host hiccup takes)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (profiling runs)")
    ap.add_argument("--only", default=None, help="profiling runs: just one row of the deep block (" + ", ".join(DEEP_ROWS) + ")")
    args = ap.parse_args()
    if args.only in ("trajectory", "layer_search", "config5_sweep"):
        torch.cuda.set_device(0)
        block = (trajectory_block(ising_operator(N_QUBITS, 2020), search_precision=os.environ.get("QSV_BENCH_SEARCH_PRECISION")) if args.only == "trajectory" else
                 layer_search_block() if args.only == "layer_search" else config5_sweep_block())
        print(json.dumps({args.only: block}), flush=True)
        return
    if args.only is not None:
        # (rocprofv3 then sees the kernels of that row alone; N = 1)
        if args.only not in DEEP_ROWS:
            raise SystemExit(f"--only takes one of {list(DEEP_ROWS) + ['trajectory', 'layer_search', 'config5_sweep']}")
        torch.cuda.set_device(0)
        print(json.dumps({"deep": deep_block(0, args.only, kernels_only=True)}), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # Rehearsal on a box with fewer GPUs than ranks (QSV_BENCH_REHEARSE=1): every rank uses the visible GPU
    # local_rank % device_count, so RCCL (one rank per device) is out and the collectives run over gloo on CPU tensors.
    # The driver's multi-GPU runs never set it.
    rehearse = os.environ.get("QSV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    comm_device = "cpu" if (rehearse and world > 1) else "cuda"

    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.distributed import evaluate_population_sharded, shard_bounds
    from queasars_amd.evqe import EVQEPopulation

    # ---- synthetic workload: one population of 64 * N individuals, rank r owns block r -------------------
    population = EVQEPopulation.random_population(N_QUBITS, N_LAYERS, POP_PER_GPU * world, True, 0)
    circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
    params = [list(ind.parameter_values) for ind in population.individuals]
    lo, hi = shard_bounds(len(circuits), world, rank)
    operator = ising_operator(N_QUBITS, 2020)
    evaluator = OperatorCircuitEvaluator(operator, device=local_rank)
    device = evaluator.statevector_device

    # The populations, circuits and plans built so far are long-lived: out of the garbage collector's way (what a
    # long-running service does after start-up), so that a full collection inside a timed region of a few milliseconds
    # does not walk them.
    gc.collect()
    gc.freeze()

    # The step's inputs are resident in HBM when the timed region starts, as the bench contract has it: the circuit structures
    # (registered: their plans are in the device arena) and the population's parameter values, one row per individual of a
    # float64 matrix in device memory (rows padded to the longest vector; an individual takes the first num_parameters values
    # of its row) -- what an optimiser that runs on the device hands over (qsv_eval_push_device).  The same step fed with host
    # lists (packed and read over PCIe inside the step) is timed the same way afterwards: `value_host_lists`.
    width = max(len(p) for p in params)
    host_matrix = np.zeros((len(params), width))
    for i, p in enumerate(params):
        host_matrix[i, : len(p)] = p
    param_matrix = torch.from_numpy(host_matrix).cuda()
    torch.cuda.synchronize()
    feed = {"values": param_matrix if os.environ.get("QSV_BENCH_HOST_LISTS") != "1" else params}

    def step():
        # the product's sharding function: this rank's block on its GPU, then one all-gather of the fitness values
        return evaluate_population_sharded(evaluator, circuits, feed["values"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Steady state before the clock starts: a fresh GPU needs tens of milliseconds of load before it holds its clocks
    # (measured: 20 timed steps right after 5 warm-up steps run 10 % slower than the same 20 steps after 0.25 s of
    # load), and 20 steps of this workload last 7 ms.  Untimed, the same step, reported in the output line.
    # (every rank must make the same number of steps -- each is a collective -- so the count is rank 0's estimate)
    t_pre = time.perf_counter()
    for _ in range(5):
        step()
    prewarm_steps = agreed_count(int(PREWARM_S / max((time.perf_counter() - t_pre) / 5, 1e-6)), world, comm_device)
    for _ in range(prewarm_steps):
        step()
    prewarm_steps += 5
    # ---- timed region: exactly K steps (no profiling events inside), barrier + synchronize on both sides, MAX over ranks.
    # A window of K steps of this workload lasts K * 60-90 us: with the driver's K = 20 a single host hiccup is a tenth
    # of it.  So the K-step window is REPEATED until at least 50 ms have been timed (every window bracketed the same way;
    # the count is rank 0's, agreed between the ranks) and the line reports the MEDIAN window; `steps` stays K.
    def window():
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    first, values = window()
    # (QSV_BENCH_WINDOWS caps the count: profiling runs, where every launch is traced, set it to 1)
    most = max(1, int(os.environ.get("QSV_BENCH_WINDOWS", 256)))
    n_windows = 1 + agreed_count(min(most - 1, int(0.05 / max(first, 1e-6))), world, comm_device)
    windows = [first]
    for _ in range(n_windows - 1):
        dt, values = window()
        windows.append(dt)
    elapsed = sorted(windows)[len(windows) // 2]
    # ... and the same K-step windows with the parameter values as host lists (list[list[float]], the reference's argument
    # type: packed into pinned memory by the host and fetched over PCIe by the kernels, all inside the step)
    host_windows = []
    inputs_agree = None
    if feed["values"] is not params:
        feed["values"] = params
        for _ in range(5):
            step()
        for _ in range(max(3, min(n_windows, 32))):
            dt, host_values = window()
            host_windows.append(dt)
        feed["values"] = param_matrix
        inputs_agree = list(host_values) == list(values)
    elapsed_host = sorted(host_windows)[len(host_windows) // 2] if host_windows else None
    # ... and with FRESH inputs every step: two device matrices taking turns (an optimiser never presents the same matrix twice;
    # the evaluator recognises a matrix it has just read and skips its checks, worth about 0.5 us)
    elapsed_fresh = None
    if feed["values"] is param_matrix:
        other = torch.from_numpy(host_matrix * 0.5 + 0.25).cuda()
        torch.cuda.synchronize()
        turn = {"k": 0}

        def step_fresh():
            turn["k"] ^= 1
            return evaluate_population_sharded(evaluator, circuits, other if turn["k"] else param_matrix)

        for _ in range(6):
            step_fresh()
        fresh_windows = []
        for _ in range(max(3, min(n_windows, 32))):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_fresh()
            fence()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=comm_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            fresh_windows.append(dt)
        elapsed_fresh = sorted(fresh_windows)[len(fresh_windows) // 2]

    # ---- after the timed region: per-kernel roofline of the same step -------------------------------------
    traffic = load_traffic("headline")
    kernels, prof = kernel_rooflines(device, step, min(args.steps, 10), traffic)
    config3 = config3_deep = None
    if not args.no_extras:
        # (the headline's handle is not needed any more, and two handles alive share the process's four hardware queues:
        # config 3's two lanes then run one after the other, 0.34 ms per step against 0.22 -- DESIGN.md section 5)
        device.close()
        config3 = config3_block(world, rank, local_rank, steps=40)
        config3_deep = config3_block(world, rank, local_rank, steps=3, layers=8)

    if rank == 0:
        total_evals = POP_PER_GPU * world * args.steps
        dominant = max(kernels, key=lambda k: k["launches"] * k["avg_launch_us"]) if kernels else None
        roofline = None
        if dominant is not None:
            roofline = {
                # what the counters say limits the dominant kernel ("latency/issue": neither roof is near); `peak` and
                # `frac` below are against the HBM roof whatever the limiter, as SURVEY 8(d) asks
                "bound": dominant["bound"],
                "roof": "hbm",
                "kernel": dominant["kernel"],
                "achieved": dominant["achieved_GBps"],
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": dominant["frac_hbm_algorithmic"],
                "traffic": dominant["traffic_bytes_per_launch"],
                "traffic_source": ("committed profile, not measured in this run: " + traffic["source"]) if traffic.get("source") else None,
                "avg_launch_us": dominant["avg_launch_us"],
                "algorithmic_bytes_per_launch": dominant["algorithmic_bytes_per_launch"],
                "observed_limiter": dominant["bound"],
                "frac_hbm_on_measured_traffic": dominant["frac_hbm_measured_traffic"],
                "frac_fp64": dominant["frac_fp64"],
                "kernels": kernels,
                "gates_per_s": prof["n_gates"] / (prof["pass_window_ms"] * 1e-3) if prof["pass_window_ms"] > 0 else None,
                "pass_window_ms_per_step": prof["pass_window_ms"] / min(args.steps, 10),
                "note": "top-level fields describe the kernel with the most GPU time.  Gate-pass kernels: achieved = "
                "algorithmic state bytes per launch (16 * 2^n per state and direction the pass has to move: pass 0 "
                "synthesises and only writes, a last pass with the fused diagonal expectation only reads, passes in "
                "between do both; SURVEY 8(d)) / mean launch time; for the virtual circuits of a split evaluation (weakly "
                "entangled circuits run as two small circuits of about n / 2 qubits, csrc/split.hpp) the states they write, "
                "2 * 16 * 2^(n/2 + keys) bytes per evaluation -- those launches are a chain of latencies (parameters over "
                "PCIe, one gate pass in LDS), nowhere near a roof, and under a quadratic diagonal operator (Ising / QUBO) "
                "the expectation value follows from weighted Gram matrices of the two small states, so NO kernel of the "
                "step sweeps 2^n amplitudes any more: the HBM roofline of the gate-pass kernel proper is roofline.microbench "
                "(single-gate sweeps) and the deep-circuit rows of DESIGN.md.  Contraction kernel (split evaluations under "
                "other diagonal operators: "
                "two small virtual circuits + one sweep, csrc/split.hpp): algorithmic bytes = what the sweep reads per "
                "state, 8 * 2^n of the diagonal table + the two side tables; the table is shared by the evaluations of a "
                "launch and each part of it is handled by one XCD, so it is read from memory once per launch and "
                "served from L2 afterwards (traffic << algorithmic bytes: observed_limiter says what the counters say).  "
                "Launch times are HIP events around every launch on its own stream, in profiled steps after the timed "
                "region.  traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 --pmc passes of this "
                "command (profiles/, see traffic_source), gfx950 correction of MI355X_MICROARCH.md",
            }
        result = {
            "metric": "circuit-evals/sec (EVQE population) at n qubits; achieved HBM GB/s vs roofline",
            "value": total_evals / elapsed,
            "unit": "circuit-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm": {"seconds": PREWARM_S, "untimed_steps": prewarm_steps,
                        "why": "GPU clocks reach steady state only after tens of ms of load; the timed K steps follow"},
            "timed_windows": {"count": len(windows), "steps_each": args.steps, "reported": "median",
                              "ms_min": min(windows) * 1e3, "ms_median": elapsed * 1e3, "ms_max": max(windows) * 1e3,
                              "why": "K steps of this workload last a millisecond or two: the K-step window is repeated until "
                                     ">= 50 ms have been timed and the median window is the one reported"},
            "ms_per_step": elapsed / args.steps * 1e3,
            "value_host_lists": (total_evals / elapsed_host) if elapsed_host else None,
            "ms_per_step_host_lists": (elapsed_host / args.steps * 1e3) if elapsed_host else None,
            "host_lists_results_identical": inputs_agree,
            "value_fresh_inputs": (total_evals / elapsed_fresh) if elapsed_fresh else None,
            "ms_per_step_fresh_inputs": (elapsed_fresh / args.steps * 1e3) if elapsed_fresh else None,
            "inputs": ("`value`: circuit structures registered (plans in the device arena) and the population's parameter values "
                       "resident in HBM before the timed region (a float64 matrix in device memory, one row per individual: "
                       "qsv_eval_push_device); results to the host.  `value_host_lists`: the same steps with the parameter "
                       "values as Python lists of floats, packed by the host and read over PCIe inside every step (what "
                       "rounds 1-2 and BENCH_r02 reported as `value`); host_lists_results_identical: the two gave the same bits") if elapsed_host else
                      "parameter values as host lists (QSV_BENCH_HOST_LISTS=1)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{N_QUBITS}-qubit EVQE population={POP_PER_GPU} per GPU, {N_LAYERS} layers, random Ising "
                f"Hamiltonian ({len(operator)} terms), fp64 (BASELINE.json configs[1])",
                "n_qubits": N_QUBITS,
                "population_per_gpu": POP_PER_GPU,
                "layers": N_LAYERS,
                "pauli_terms": len(operator),
                "parallelism": (f"population sharded over {world} GPU(s); fitness values exchanged through a table in shared host memory the "
                                "kernels store into (one node; RCCL all-gather: fallback and first-step check)") if world > 1 else "1 GPU",
                "path": "library defaults (DESIGN.md 4.2 / 5): register splitting with up to five cut keys, one launch per push "
                        "where both virtual circuits are one tile, factorised Ising expectation, multiplexed gates, chain "
                        "stream, end of a batch read off the pinned result buffer, a repeated batch keeps its layout",
            },
            "roofline": roofline,
        }
        if config3 is not None:
            result["config3"] = config3
            result["config3_deep"] = config3_deep
        if world == 1 and not args.no_extras:
            result["trajectory"] = trajectory_block(operator)
            fp32_search = trajectory_block(operator, search_precision="fp32")
            result["trajectory"]["with_fp32_search_evaluator"] = {k: fp32_search[k] for k in ("evaluations", "seconds", "evals_per_s", "eigenvalue")}
            result["layer_search"] = layer_search_block()
            result["config5_sweep"] = config5_sweep_block()
            result["deep"] = deep_block(local_rank)
        if world == 1 and not args.no_extras:
            result["roofline"]["microbench"] = microbench_block()
            cold, threaded, noop, blocking = cold_and_threaded(operator)
            result["cold_structure_evals_per_s"] = cold
            result["threaded_b1_evals_per_s"] = threaded
            result["threaded_b1_noop_tasks_per_s"] = noop
            result["threaded_b1_blocking_tasks_per_s"] = blocking
            result["sampler_branch"] = sampler_block(operator, circuits, params)
            result["parameter_search"] = search_block(operator, population)
            result["calling_pattern_note"] = (
                "cold: every step evaluates 64 circuit structures the device has never seen (plan building + upload "
                "inside the timed region); threaded: 64 host threads, one circuit per call (the reference's selection "
                "operator, selection.py:75-82) through CoalescingCircuitEvaluator; threaded_b1_noop_tasks_per_s: the same "
                "pool.map with tasks that do nothing, i.e. the rate at which this host's CPython can hand out and collect "
                "such calls at all; threaded_b1_blocking_tasks_per_s: the same with tasks that sleep 0.1 ms outside the "
                "interpreter -- what 64 threads that each BLOCK once per task can reach, whatever they wait for")
        if world == 1 and not args.no_cpu_baseline:
            base, ref_values = cpu_baseline(circuits, params, operator, gpu_values=values)
            result["cpu_baseline"] = base
            err = float(np.abs(np.asarray(values[: len(ref_values)]) - np.asarray(ref_values)).max())
            result["max_abs_diff_vs_cpu_oracle"] = err
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
