"""Population sharding over the GPUs of one node: one process per GPU, one fitness all-gather per evaluation.

The reference farms one task per individual to a thread pool or a Dask cluster
(queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/selection.py:75-85, mutation.py:206-218).  Here the
individuals are split into contiguous blocks by population index; each rank evaluates its block on its own GPU
with no communication, and a single all-gather of ``P / world`` doubles per rank (``backend="nccl"`` is RCCL on
ROCm) gives every rank all P fitness values, which is what selection needs (selection.py:85, :102).  The
collective moves a few hundred bytes: it is latency bound, xGMI bandwidth does not matter.
"""

from __future__ import annotations

import os
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


def _rows(parameter_values, lo: int, hi: int):
    """Rows [lo, hi) of the population's parameter values: a list of vectors, or -- for a matrix that lives in device memory
    (a torch tensor; ``OperatorCircuitEvaluator.evaluate_circuits`` reads it where it is) -- a view of its rows."""
    if getattr(parameter_values, "is_cuda", False):
        return parameter_values if lo == 0 and hi == len(parameter_values) else parameter_values[lo:hi]
    return list(parameter_values[lo:hi])


def evaluate_population_sharded(evaluator, circuits: Sequence, parameter_values: Sequence, group=None, device=None) -> list[float]:
    """Evaluate this rank's block of the population and all-gather the fitness values.

    ``evaluator`` is any object with ``evaluate_circuits(circuits, parameter_values)``; every rank must pass the
    same full ``circuits`` / ``parameter_values`` lists (or, for evaluators that take it, the population's parameter matrix
    in the rank's own device memory).  Returns all values, ordered by population index, on
    every rank.  Without an initialised process group (or with world size 1) it evaluates everything locally.
    """
    import torch
    import torch.distributed as dist

    n = len(circuits)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(evaluator.evaluate_circuits(list(circuits), _rows(parameter_values, 0, n)))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    if device is None and dist.get_backend(group) == "nccl" and os.environ.get("QSV_GATHER_CHAIN", "1") != "0":
        # evaluation, collective and copy back as one chain on one stream (no host round trip in between)
        chained = evaluate_block_and_gather(evaluator, circuits, parameter_values, n, world, rank, group,
                                           torch.device("cuda", torch.cuda.current_device()))
        if chained is not None:
            return chained
    local = evaluator.evaluate_circuits(list(circuits[lo:hi]), _rows(parameter_values, lo, hi)) if hi > lo else []
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    owner = id(getattr(evaluator, "statevector_device", None) or evaluator)
    return _gather(local, n, world, rank, group, torch.device(device), owner)


def evaluate_block_and_gather(evaluator, circuits: Sequence, parameter_values: Sequence, n: int, world: int, rank: int, group, device):
    """This rank's block and the all-gather as ONE chain on one HIP stream, with one synchronisation at its end: the
    evaluator leaves its values in the collective's send buffer (``evaluate_circuits_to_device``: device memory, no wait),
    the all-gather and the copy to the host follow on the same stream.  Measured on one MI355X (nccl group of one rank,
    ``scripts/gatherstep.py``): the step with the gather costs 138 us the old way (evaluate, wait, stage, copy to the
    device, gather, copy back, wait) against 91 us for the evaluation alone.  Returns None when the evaluator cannot leave
    its values on the device (the caller then takes the old way).

    The chain runs on a stream of its own (the one the evaluator's device launches on).  It is ordered behind whatever the
    caller has queued on ITS current stream (one event wait), and the caller's current stream is the same after the call
    as before it: torch work of the host application never ends up on the chain's stream.

    The first chained step of an evaluator in a group is also run the staged way and the two results compared: a
    difference (an RCCL build or a stream ordering this path was never run on) switches that evaluator back to the staged
    path for good, with a warning, instead of handing selection wrong fitness values."""
    import torch
    import torch.distributed as dist

    to_device = getattr(evaluator, "evaluate_circuits_to_device", None)
    if to_device is None or device.type != "cuda":
        return None
    state = _chain_state(evaluator, device)
    if state is None or state.get("disabled"):
        return None
    lo, hi = shard_bounds(n, world, rank)
    width = -(-n // world)
    _, send, recv, recv_host = _buffers(world, width, device, group, state["key"])
    stream = state["stream"]
    caller = torch.cuda.current_stream(device)
    if caller != stream:
        stream.wait_stream(caller)  # (the application's earlier work on its own stream comes first)
        torch.cuda.set_stream(stream)
    try:
        if hi - lo < width:
            send.fill_(float("nan"))
        if hi > lo and not to_device(circuits[lo:hi], _rows(parameter_values, lo, hi), send.data_ptr()):
            return None
        dist.all_gather_into_tensor(recv, send, group=group)
        recv_host.copy_(recv, non_blocking=True)
        stream.synchronize()
    finally:
        if caller != stream:
            torch.cuda.set_stream(caller)
    values = _unpack(recv_host, n, world, width)
    verified = state["verified"]
    if id(group) not in verified:
        local = evaluator.evaluate_circuits(list(circuits[lo:hi]), _rows(parameter_values, lo, hi)) if hi > lo else []
        staged = _gather(local, n, world, rank, group, device, state["key"])
        same = len(staged) == len(values) and all(a == b for a, b in zip(staged, values))
        # every rank must take the same path from now on: agree on the verdict (a collective itself, staged way)
        flag = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag.item()) != 1.0:
            import warnings

            warnings.warn("queasars_amd.distributed: the chained evaluate + all-gather step disagreed with the staged one; "
                          "using the staged path", RuntimeWarning)
            state["disabled"] = True
            return staged
        verified.add(id(group))
    return values


def _unpack(recv_host, n: int, world: int, width: int) -> list[float]:
    """The gathered slots (one of ``width`` values per rank, the unused tail NaN) as one list ordered by population index."""
    table = recv_host.numpy().reshape(world, width)
    if n == world * width:
        return table.ravel().tolist()
    out: list[float] = []
    for r in range(world):
        rlo, rhi = shard_bounds(n, world, r)
        out.extend(table[r, : rhi - rlo].tolist())
    return out


_CHAIN_STATE: dict = {}


def _chain_state(evaluator, device):
    """Per evaluator device: the HIP stream it launches on as a torch stream (created once and handed to the library: its
    kernels, the collective and the copy back are then ordered by the stream alone), which groups the chained step has
    been verified on, and the key of its buffers.  The entry goes when the device does."""
    import torch
    import weakref

    dev = getattr(evaluator, "statevector_device", None)
    if dev is None:
        return None
    hit = _CHAIN_STATE.get(id(dev))
    if hit is None or hit["ref"]() is not dev:
        key = id(dev)

        def gone(_ref, key=key):
            _CHAIN_STATE.pop(key, None)
            for k in [k for k in _BUFFERS if k[-1] == key]:
                _BUFFERS.pop(k, None)

        stream = torch.cuda.Stream(device=device)
        dev.set_stream(stream.cuda_stream)
        hit = {"ref": weakref.ref(dev, gone), "stream": stream, "verified": set(), "disabled": False, "key": key}
        _CHAIN_STATE[key] = hit
    return hit


def _gather(local: Sequence[float], n: int, world: int, rank: int, group, device, owner=None) -> list[float]:
    """All ranks' blocks of fitness values, ordered by population index (``local`` is this rank's block)."""
    import torch
    import torch.distributed as dist

    width = -(-n // world)  # every rank contributes a fixed-size slot so one all_gather_into_tensor suffices
    send_host, send, recv, recv_host = _buffers(world, width, device, group, owner)
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
    return _unpack(recv_host, n, world, width)


_BUFFERS: dict = {}


def _buffers(world: int, width: int, device, group=None, owner=None):
    """(pinned send staging, device send, device receive, pinned receive staging) for one (world, width, device, group,
    calling thread, owner): two evaluators, groups or threads never share a send / receive buffer (``owner`` = the
    evaluator device of a chained step; its entries are dropped when it dies)."""
    import threading

    import torch

    key = (world, width, str(device), id(group) if group is not None else 0, threading.get_ident(), owner)
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
