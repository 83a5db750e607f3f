"""Population sharding + fitness all-gather over two CPU ranks (gloo).  The evaluator is mocked with the oracle:
what is under test is the N > 1 control path of queasars_amd.distributed, not the GPU kernels."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleEvaluator:
    """Stands in for the GPU evaluator on CPU ranks; records which individuals this rank evaluated."""

    def __init__(self, operator):
        self.operator = operator
        self.seen = 0

    def evaluate_circuits(self, circuits, parameter_values):
        import helpers

        self.seen += len(circuits)
        return [helpers.oracle_expectation(c, p, self.operator) for c, p in zip(circuits, parameter_values)]


def _worker(rank: int, world: int, port: int, n_individuals: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import helpers
    from queasars_amd.distributed import evaluate_population_sharded, shard_bounds

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, circuits, params = helpers.population_circuits(5, 2, n_individuals, seed=42)
        op = helpers.random_ising_operator(5, seed=7)
        evaluator = _OracleEvaluator(op)
        values = evaluate_population_sharded(evaluator, circuits, params)
        lo, hi = shard_bounds(n_individuals, world, rank)
        assert evaluator.seen == hi - lo, "a rank must only evaluate its own block"
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.asarray(values))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_individuals", [8, 7, 1])
def test_two_rank_sharded_evaluation(tmp_path, n_individuals):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_individuals, str(tmp_path)), nprocs=world, join=True)
    import helpers

    _, circuits, params = helpers.population_circuits(5, 2, n_individuals, seed=42)
    op = helpers.random_ising_operator(5, seed=7)
    want = np.asarray([helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)])
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npy")
        assert got.shape == want.shape and np.array_equal(got, want), "every rank must hold all fitness values in order"


def test_shard_bounds_cover_everything_once():
    from queasars_amd.distributed import shard_bounds

    for n in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_single_process_falls_back_to_local_evaluation():
    import helpers
    from queasars_amd.distributed import evaluate_population_sharded

    _, circuits, params = helpers.population_circuits(4, 2, 3, seed=1)
    op = helpers.random_ising_operator(4, seed=2)
    ev = _OracleEvaluator(op)
    values = evaluate_population_sharded(ev, circuits, params)
    assert ev.seen == 3 and len(values) == 3


# ---- the same path on RCCL, when the box has two GPUs ----------------------------------------------------------------


def _nccl_worker(rank: int, world: int, port: int, n_individuals: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import helpers
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
    from queasars_amd.distributed import evaluate_population_sharded

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        n = 12
        _, circuits, params = helpers.population_circuits(n, 3, n_individuals, seed=42)
        evaluator = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=7), device=rank)
        values = evaluate_population_sharded(evaluator, circuits, params)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.asarray(values))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_evaluation_on_rccl(tmp_path):
    """World size 2 over RCCL (backend "nccl"), one process per GPU: every rank ends with all fitness values, equal
    to a single-GPU evaluation bit for bit.  Skipped on a one-GPU box."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world, n_individuals = 2, 9
    mp.spawn(_nccl_worker, args=(world, _free_port(), n_individuals, str(tmp_path)), nprocs=world, join=True)
    import helpers
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

    _, circuits, params = helpers.population_circuits(12, 3, n_individuals, seed=42)
    want = np.asarray(OperatorCircuitEvaluator(helpers.random_ising_operator(12, seed=7)).evaluate_circuits(circuits, params))
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npy")
        assert np.array_equal(got, want)


def test_gather_buffers_are_not_shared_between_evaluators_groups_or_threads():
    """Advisor finding (round 2): the staging buffers were keyed by (world, width, device) alone, so two evaluators (or
    two threads) with the same shard width overwrote each other's fitness values."""
    import threading

    import torch

    from queasars_amd.distributed import _BUFFERS, _buffers

    cpu = torch.device("cpu")
    a = _buffers(2, 4, cpu, None, 101)
    assert _buffers(2, 4, cpu, None, 101)[0] is a[0], "the same caller gets its buffers back"
    b = _buffers(2, 4, cpu, None, 202)
    assert b[0] is not a[0] and b[3] is not a[3]
    group = object()
    assert _buffers(2, 4, cpu, group, 101)[0] is not a[0]
    other = []
    t = threading.Thread(target=lambda: other.append(_buffers(2, 4, cpu, None, 101)))
    t.start()
    t.join()
    assert other[0][0] is not a[0]
    for key in [k for k in _BUFFERS if k[-1] in (101, 202)]:
        _BUFFERS.pop(key)
