#!/usr/bin/env python3
"""Headline benchmark: circuit-evals/sec of an EVQE population on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "Config 2"): n = 20 qubits, P = 64 individuals per GPU,
L = 4 layers, genomes from the restated ``EVQEPopulation.random_population(..., random_seed=0)``, random Ising
Hamiltonian (190 ZZ + 20 Z terms, J, h ~ N(0,1), default_rng(2020)), fp64.  One "step" = one fitness evaluation
of the rank's 64 individuals (``evaluate_circuits`` on all of them) followed by the fitness all-gather.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Weak scaling: every rank evaluates its own 64 individuals (rank r takes individuals [64r, 64r+64) of one
64*N population), so value = 64 * N * K / T.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
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
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def ising_operator(n_qubits: int, seed: int):
    from queasars_amd.ir import PauliOperator

    rng = np.random.default_rng(seed)
    terms = []
    for i in range(n_qubits):
        for j in range(i + 1, n_qubits):
            terms.append(("ZZ", [i, j], float(rng.normal())))
    for i in range(n_qubits):
        terms.append(("Z", [i], float(rng.normal())))
    return PauliOperator.from_sparse_list(terms, n_qubits)


def cpu_baseline(circuits, params, operator, budget_s: float = 12.0):
    """Time the plain-C oracle (OpenMP, all host cores) on a bounded sample of the same workload."""
    import helpers  # tests/helpers.py: the only place outside tests/ that touches oracle/, as the timed baseline

    orc = helpers.load_c_oracle()
    # the GPU box hands one job a 16-core share of the host: do not oversubscribe it
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    orc.lib.qsvo_set_threads(max(1, min(share, 16)))
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
    return {
        "value": done / elapsed,
        "unit": "circuit-evals/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {done} of the {len(circuits)} individuals of the same workload, plain-C oracle with OpenMP "
        f"({cores} threads), one sweep per gate, diagonal table prebuilt",
    }, values


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.evqe import EVQEPopulation

    # ---- synthetic workload: one population of 64 * N individuals, rank r owns block r -------------------
    population = EVQEPopulation.random_population(N_QUBITS, N_LAYERS, POP_PER_GPU * world, True, 0)
    mine = population.individuals[rank * POP_PER_GPU : (rank + 1) * POP_PER_GPU]
    circuits = [ind.get_parameterized_quantum_circuit() for ind in mine]
    params = [list(ind.parameter_values) for ind in mine]
    operator = ising_operator(N_QUBITS, 2020)
    evaluator = OperatorCircuitEvaluator(operator, device=local_rank)
    device = evaluator.statevector_device
    fitness_all = torch.empty(world * POP_PER_GPU, dtype=torch.float64, device="cuda")

    def step():
        local = evaluator.evaluate_circuits(circuits, params)
        if world > 1:
            send = torch.as_tensor(np.asarray(local), device="cuda")
            dist.all_gather_into_tensor(fitness_all, send)  # RCCL: 64 doubles per rank
        return local

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps -------------------------------------------------------------------
    device.set_profiling(True)  # HIP events on the library's stream around every group of pass launches
    prof = {"pass_ms": 0.0, "pass_window_ms": 0.0, "n_pass_launches": 0, "state_bytes": 0, "moved_bytes": 0, "n_state_passes": 0,
            "expect_ms": 0.0, "n_gates": 0}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        values = step()
        p = device.profile()
        for key in prof:
            prof[key] += p[key]
    fence()
    elapsed = time.perf_counter() - t0
    device.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_evals = POP_PER_GPU * world * args.steps
        launches = max(prof["n_pass_launches"], 1)
        avg_launch_ms = prof["pass_ms"] / launches
        bytes_per_launch = prof["state_bytes"] / launches
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic = None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get("pass_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "circuit-evals/sec (EVQE population) at n qubits; achieved HBM GB/s vs roofline",
            "value": total_evals / elapsed,
            "unit": "circuit-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
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
                "parallelism": f"population sharded over {world} GPU(s), RCCL all-gather of fitness" if world > 1 else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "qsv::pass_kernel<double, 3, 2, *> (both instantiations: the synthesising pass 0 and later passes)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "launches": launches,
                "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "gates_per_s": prof["n_gates"] / (prof["pass_window_ms"] * 1e-3) if prof["pass_window_ms"] > 0 else None,
                # launches of different pushes overlap on two HIP streams: the same bytes over the wall-clock window
                # of a step's passes is the chip-level rate
                "aggregate_achieved": prof["state_bytes"] / (prof["pass_window_ms"] * 1e-3) / 1e9
                if prof["pass_window_ms"] > 0 else None,
                "pass_window_ms_per_step": prof["pass_window_ms"] / args.steps,
                # state bytes the launches really moved (a compact first pass replaces the state round trip between
                # the first two passes by a small table) and, for comparison with sweep-per-gate simulators, the same
                # launches at SURVEY.md 8(d)'s flat price of 32 * 2^n bytes per state and pass
                "moved_bytes_per_launch": prof["moved_bytes"] / launches,
                "sweep_equivalent_achieved": prof["n_state_passes"] * 32.0 * (1 << N_QUBITS) / (prof["pass_ms"] * 1e-3) / 1e9
                if prof["pass_ms"] > 0 else None,
                "note": "achieved = algorithmic state bytes per pass launch / mean launch time (HIP events on the stream "
                "each launch runs on).  Algorithmic bytes: 16 * 2^n per state and direction a fused-pass design has to "
                "move (SURVEY 8(d): 32 * 2^n per full pass; pass 0 synthesises and only writes, the last pass fuses the "
                "expectation and only reads).  The compact first pass (DESIGN.md 4.1 item 1b) then avoids most of that "
                "traffic (moved_bytes_per_launch, and `traffic` from the PMC counters), so the passes are bound by fp64 "
                "issue, LDS traffic and latency rather than by HBM",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            base, ref_values = cpu_baseline(circuits, params, operator)
            result["cpu_baseline"] = base
            err = float(np.abs(np.asarray(values[: len(ref_values)]) - np.asarray(ref_values)).max())
            result["max_abs_diff_vs_cpu_oracle"] = err
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
