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


# ---- cost-aware shares ("sort by gate count and deal round-robin if imbalance > 10 %") --------------------


def test_partition_by_cost():
    from queasars_amd.distributed import contiguous_shares, imbalance, partition_by_cost

    # even costs: the contiguous blocks stay
    assert partition_by_cost([1.0] * 64, 8) == contiguous_shares(64, 8)
    assert partition_by_cost([1.0, 1.05, 0.95, 1.0], 2) == [[0, 1], [2, 3]]
    # one deep individual among shallow ones (17 us against 1 us): its block would be 2.4 times the mean
    costs = [1.0] * 16
    costs[3] = 17.0
    blocks = contiguous_shares(16, 4)
    assert imbalance(blocks, costs) > 2.0
    shares = partition_by_cost(costs, 4)
    assert sorted(i for share in shares for i in share) == list(range(16))
    assert [3] in shares and imbalance(shares, costs) < imbalance(blocks, costs)  # (the deep one alone: it IS the longest share)
    # a mixed-depth population: within 10 % of even
    rng = np.random.default_rng(0)
    costs = list(np.where(rng.random(64) < 0.3, 17.0, 1.0) * rng.uniform(0.9, 1.1, 64))
    for world in (2, 4, 8):
        shares = partition_by_cost(costs, world)
        assert sorted(i for share in shares for i in share) == list(range(64))
        assert imbalance(shares, costs) <= 1.10, (world, imbalance(shares, costs))
        assert shares == partition_by_cost(costs, world)  # (a pure function of the costs)
        assert all(share == sorted(share) for share in shares)
    assert partition_by_cost([], 4) == [[], [], [], []]
    assert partition_by_cost([3.0, 1.0], 1) == [[0, 1]]


class _CostedOracleEvaluator(_OracleEvaluator):
    """... that also tells what a circuit costs (as OperatorCircuitEvaluator.circuit_costs does): here by its depth."""

    def __init__(self, operator):
        super().__init__(operator)
        self.seen_circuits = []

    def evaluate_circuits(self, circuits, parameter_values):
        self.seen_circuits.extend(circuits)
        return super().evaluate_circuits(circuits, parameter_values)

    @property
    def n_qubits(self):
        return self.operator.num_qubits

    def circuit_costs(self, circuits):
        return [{"microseconds": 17.0 if c.depth() >= 8 else 1.0} for c in circuits]


def _mixed_population(n_individuals: int):
    """Four-layer individuals with one eight-layer individual among them."""
    import helpers
    from queasars_amd.evqe import EVQEPopulation

    shallow = EVQEPopulation.random_population(5, 4, n_individuals, True, 3).individuals
    deep = EVQEPopulation.random_population(5, 8, 1, True, 4).individuals
    individuals = list(shallow)
    individuals[1] = deep[0]
    circuits = [ind.get_parameterized_quantum_circuit() for ind in individuals]
    params = [list(ind.parameter_values) for ind in individuals]
    return individuals, circuits, params, helpers.random_ising_operator(5, seed=7)


def _balanced_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from queasars_amd.distributed import evaluate_population_sharded, imbalance, population_shares

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, circuits, params, op = _mixed_population(20)
        evaluator = _CostedOracleEvaluator(op)
        values = evaluate_population_sharded(evaluator, circuits, params)
        shares = population_shares(evaluator, circuits, world)
        costs = [c["microseconds"] for c in evaluator.circuit_costs(circuits)]
        assert imbalance(shares, costs) <= 1.10
        assert [id(c) for c in evaluator.seen_circuits] == [id(circuits[i]) for i in shares[rank]], "a rank evaluates its share only"
        again = evaluate_population_sharded(evaluator, circuits, params)  # (the shares are remembered: the same answer)
        assert again == values
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.asarray(values))
        np.save(os.path.join(out_dir, f"share{rank}.npy"), np.asarray(shares[rank]))
    finally:
        dist.destroy_process_group()


def test_two_rank_evaluation_with_one_deep_individual_among_shallow_ones(tmp_path):
    """One eight-layer circuit among four-layer ones: the shares are dealt by cost (within 10 % of even, where the contiguous
    blocks are 26 : 10, the dealt shares 18 : 18 -- the deep individual and one shallow one against eighteen shallow ones), and the values come back in population order, identical to the unsharded call."""
    world = 2
    mp.spawn(_balanced_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import helpers

    _, circuits, params, op = _mixed_population(20)
    want = np.asarray([helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)])
    shares = [np.load(tmp_path / f"share{r}.npy").tolist() for r in range(world)]
    assert sorted(shares[0] + shares[1]) == list(range(20)) and sorted(map(len, shares)) == [2, 18]
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"rank{rank}.npy"), want)


def _search_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        result, seen = _run_small_evolution(shard=True)
        np.save(os.path.join(out_dir, f"values{rank}.npy"), np.asarray(result.best_expectation_values + [result.eigenvalue]))
        np.save(os.path.join(out_dir, f"best{rank}.npy"), np.asarray(result.best_individual.parameter_values))
        np.save(os.path.join(out_dir, f"counts{rank}.npy"), np.asarray(result.circuit_evaluations + [seen]))
    finally:
        dist.destroy_process_group()


def _run_small_evolution(shard: bool):
    import helpers
    from queasars_amd.evqe.solver import SPSA, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration

    op = helpers.random_ising_operator(4, seed=5)
    evaluator = _CostedOracleEvaluator(op)
    cfg = EVQEMinimumEigensolverConfiguration(
        optimizer=SPSA(maxiter=6, learning_rate=0.4, perturbation=0.3), population_size=7, max_generations=3, random_seed=1,
        n_initial_layers=2, randomize_initial_population_parameters=True, speciation_genetic_distance_threshold=2,
        use_tournament_selection=True, tournament_size=2, selection_alpha_penalty=0.1, selection_beta_penalty=0.1,
        parameter_search_probability=0.5, topological_search_probability=0.5, layer_removal_probability=0.05)
    result = EVQEMinimumEigensolver(cfg, shard=shard).compute_minimum_eigenvalue(evaluator)
    return result, evaluator.seen


def test_two_rank_sharded_searches_are_the_single_rank_evolution(tmp_path):
    """Every rank runs the optimiser runs of ITS individuals and one all-gather per search hands everybody the results
    (mutation.py:206-218): the evolution -- best values per generation, the best individual's iterate, evaluation counts -- is
    the single-process one on every rank, and each rank has evaluated about half of it."""
    world = 2
    mp.spawn(_search_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    single, seen_single = _run_small_evolution(shard=False)
    want_values = np.asarray(single.best_expectation_values + [single.eigenvalue])
    seen = []
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"values{rank}.npy"), want_values)
        assert np.array_equal(np.load(tmp_path / f"best{rank}.npy"), np.asarray(single.best_individual.parameter_values))
        counts = np.load(tmp_path / f"counts{rank}.npy")
        assert counts[:-1].tolist() == single.circuit_evaluations
        seen.append(int(counts[-1]))
    assert sum(seen) == seen_single == sum(single.circuit_evaluations)
    assert max(seen) < 0.75 * seen_single


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


class _PointerEvaluator(_OracleEvaluator):
    """An evaluator that can leave its values where the caller says (``evaluate_circuits_to_device``): here the "device"
    is the host, the pointer a host address -- what the node's shared table hands a CPU rank."""

    def __init__(self, operator):
        super().__init__(operator)
        self.to_pointer = 0

    def evaluate_circuits_to_device(self, circuits, parameter_values, pointer):
        import ctypes

        values = np.asarray(self.evaluate_circuits(circuits, parameter_values), dtype=np.float64)
        ctypes.memmove(pointer, values.ctypes.data, values.nbytes)
        self.to_pointer += 1
        return True


def _node_table_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import helpers
    from queasars_amd import distributed

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        op = helpers.random_ising_operator(5, seed=7)
        table = distributed._node_table(None, world, rank, torch.device("cpu"))
        assert table is not None, "two ranks on one host share a table"
        assert not [f for f in os.listdir("/dev/shm") if f.startswith("qsv_fitness_")], "the name is gone once everybody has it open"
        got = []
        for step, n_individuals in enumerate([8, 7, 1, 8, 8, 3, 8]):  # (uneven and empty shares, both tables several times over)
            _, circuits, params = helpers.population_circuits(5, 2, n_individuals, seed=40 + step)
            evaluator = _PointerEvaluator(op) if step % 2 == 0 else _OracleEvaluator(op)
            values = distributed.evaluate_population_sharded(evaluator, circuits, params)
            lo, hi = distributed.shard_bounds(n_individuals, world, rank)
            assert evaluator.seen == hi - lo
            if step % 2 == 0 and hi > lo:
                assert evaluator.to_pointer == 1, "the values went straight into the rank's slot"
            got.append(values)
        assert table.step == 7
        # shares wider than a slot: the collective takes over, same values
        table.capacity, keep = 2, table.capacity
        _, circuits, params = helpers.population_circuits(5, 2, 8, seed=40)
        assert distributed.evaluate_population_sharded(_OracleEvaluator(op), circuits, params) == got[0]
        table.capacity = keep
        os.environ["QSV_GATHER_NODE"] = "0"
        assert distributed.evaluate_population_sharded(_OracleEvaluator(op), circuits, params) == got[0]
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate([np.asarray(v) for v in got]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_through_the_nodes_shared_table(tmp_path):
    """The all-gather without a collective: both ranks map one table in POSIX shared memory, each leaves its values in its
    slot, everybody reads the table (queasars_amd.distributed._NodeTable)."""
    world = 2
    mp.spawn(_node_table_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import helpers

    op = helpers.random_ising_operator(5, seed=7)
    want = []
    for step, n_individuals in enumerate([8, 7, 1, 8, 8, 3, 8]):
        _, circuits, params = helpers.population_circuits(5, 2, n_individuals, seed=40 + step)
        want += [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npy")
        assert got.shape == (len(want),) and np.array_equal(got, np.asarray(want)), np.abs(got - np.asarray(want)).max()


def _gpu_node_table_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import helpers
    from queasars_amd import distributed
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

    torch.cuda.set_device(0)  # (both ranks on the one GPU of the box: the table is host memory, every GPU is given it alike)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 14
        op = helpers.random_ising_operator(n, seed=7)
        evaluator = OperatorCircuitEvaluator(op, device=0)
        got = []
        for step, count in enumerate([8, 7, 1, 12, 8, 3]):
            _, circuits, params = helpers.population_circuits(n, 3, count, seed=40 + step)
            got.append(distributed.evaluate_population_sharded(evaluator, circuits, params))
        table = distributed._node_table(None, world, rank, torch.device("cuda", 0))
        assert table is not None and table.registered and table.step == 6, "the GPUs stored into the node's shared table"
        # the parameter values in the rank's own device memory, and the same population again: the same bits
        _, circuits, params = helpers.population_circuits(n, 3, 12, seed=43)
        host_matrix = np.zeros((len(params), max(len(p) for p in params)))
        for row, p in zip(host_matrix, params):
            row[: len(p)] = p
        matrix = torch.from_numpy(host_matrix).cuda()
        assert distributed.evaluate_population_sharded(evaluator, circuits, matrix) == got[3]
        os.environ["QSV_GATHER_NODE"] = "0"
        assert distributed.evaluate_population_sharded(evaluator, circuits, params) == got[3], "the collective's values, bit for bit"
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate([np.asarray(v) for v in got]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_kernels_store_into_the_nodes_shared_table(tmp_path):
    """Two ranks (gloo for the set-up, both on GPU 0) with real evaluators: each rank's kernels store its values straight into
    its slot of the node's shared table, registered with HIP; values against the oracle, and the collective's bits."""
    world = 2
    mp.spawn(_gpu_node_table_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import helpers

    op = helpers.random_ising_operator(14, seed=7)
    want = []
    for step, count in enumerate([8, 7, 1, 12, 8, 3]):
        _, circuits, params = helpers.population_circuits(14, 3, count, seed=40 + step)
        want += [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    got0, got1 = (np.load(tmp_path / f"rank{r}.npy") for r in range(world))
    assert np.array_equal(got0, got1)
    assert np.abs(got0 - np.asarray(want)).max() < 1e-10
