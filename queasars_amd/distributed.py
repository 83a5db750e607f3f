"""Population sharding over the GPUs of one node: one process per GPU, one fitness all-gather per evaluation.

The reference farms one task per individual to a thread pool or a Dask cluster
(queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/selection.py:75-85, mutation.py:206-218).  Here the
individuals are split into contiguous blocks by population index; each rank evaluates its block on its own GPU
with no communication, and a single all-gather of ``P / world`` doubles per rank (``backend="nccl"`` is RCCL on
ROCm) gives every rank all P fitness values, which is what selection needs (selection.py:85, :102).  The
collective moves a few hundred bytes: it is latency bound, xGMI bandwidth does not matter.
"""

from __future__ import annotations

from typing import Optional, Sequence

import numpy as np


def shard_bounds(n_items: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of ``n_items`` owned by ``rank``; the first ``n_items % world_size`` ranks get
    one item more."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def evaluate_population_sharded(evaluator, circuits: Sequence, parameter_values: Sequence, group=None, device=None) -> list[float]:
    """Evaluate this rank's block of the population and all-gather the fitness values.

    ``evaluator`` is any object with ``evaluate_circuits(circuits, parameter_values)``; every rank must pass the
    same full ``circuits`` / ``parameter_values`` lists.  Returns all values, ordered by population index, on
    every rank.  Without an initialised process group (or with world size 1) it evaluates everything locally.
    """
    import torch
    import torch.distributed as dist

    n = len(circuits)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(evaluator.evaluate_circuits(list(circuits), list(parameter_values)))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    local = evaluator.evaluate_circuits(list(circuits[lo:hi]), list(parameter_values[lo:hi])) if hi > lo else []
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    return _gather(local, n, world, rank, group, torch.device(device))


def _gather(local: Sequence[float], n: int, world: int, rank: int, group, device) -> list[float]:
    """All ranks' blocks of fitness values, ordered by population index (``local`` is this rank's block)."""
    import torch
    import torch.distributed as dist

    width = -(-n // world)  # every rank contributes a fixed-size slot so one all_gather_into_tensor suffices
    send_host, send, recv, recv_host = _buffers(world, width, device)
    # pinned staging buffers and device tensors are kept between calls: the collective moves a few hundred bytes and
    # is latency bound, so every allocation and every synchronous pageable copy on its path counts
    staged = send_host.numpy()
    staged[: len(local)] = local
    staged[len(local):] = np.nan
    send.copy_(send_host, non_blocking=True)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv_host.copy_(recv, non_blocking=True)
    if recv.is_cuda:
        torch.cuda.current_stream(recv.device).synchronize()
    table = recv_host.numpy().reshape(world, width)
    if n == world * width:
        return table.ravel().tolist()
    out: list[float] = []
    for r in range(world):
        rlo, rhi = shard_bounds(n, world, r)
        out.extend(table[r, : rhi - rlo].tolist())
    return out


_BUFFERS: dict = {}


def _buffers(world: int, width: int, device):
    """(pinned send staging, device send, device receive, pinned receive staging) for one (world, width, device)."""
    import torch

    key = (world, width, str(device))
    hit = _BUFFERS.get(key)
    if hit is None:
        pin = device.type == "cuda"
        send_host = torch.empty(width, dtype=torch.float64, pin_memory=pin)
        recv_host = torch.empty(world * width, dtype=torch.float64, pin_memory=pin)
        if device.type == "cuda":
            send = torch.empty(width, dtype=torch.float64, device=device)
            recv = torch.empty(world * width, dtype=torch.float64, device=device)
        else:
            send, recv = send_host, recv_host
        hit = (send_host, send, recv, recv_host)
        _BUFFERS[key] = hit
    return hit
