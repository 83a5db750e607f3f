"""Population sharding over the GPUs of one node: one process per GPU, one fitness all-gather per evaluation.

The reference farms one task per individual to a thread pool or a Dask cluster
(queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/selection.py:75-85, mutation.py:206-218): its pool balances
dynamically -- a worker that drew a shallow individual takes the next one.  Here every rank evaluates a fixed share of the
population on its own GPU with no communication, and a single all-gather of ``ceil(P / world)`` doubles per rank
(``backend="nccl"`` is RCCL on ROCm) gives every rank all P fitness values, which is what selection needs (selection.py:85,
:102).  The collective moves a few hundred bytes: it is latency bound, xGMI bandwidth does not matter.

On ONE node the all-gather does not need a collective at all (round 4): the ranks share a table of fitness slots in POSIX
shared memory that every GPU can write (``_NodeTable``); a rank's evaluation leaves its values in the rank's slot straight
from the kernel that computed them, and every rank reads the end of the step off the table -- each slot starts as a sentinel
no arithmetic produces.  The RCCL all-gather stays as the path for groups that span nodes and as the fallback
(``QSV_GATHER_NODE=0``), and the first step on a group is run both ways and compared.

Shares.  Contiguous blocks by population index while the blocks are even; once the evaluator can tell what a
circuit costs on its device (``circuit_costs``: the route a circuit takes there, ``qsv_circuit_cost``) and the blocks' costs
differ by more than 10 %, the individuals are dealt by longest processing time first -- an eight-layer individual without
a split form costs eighteen shallow ones, and one block of them would leave seven GPUs waiting.  The deal is a pure function
of the costs (every rank computes the same one); results go back to population order.
"""

from __future__ import annotations

import os
import time
import weakref
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


def contiguous_shares(n_items: int, world_size: int) -> list[list[int]]:
    return [list(range(*shard_bounds(n_items, world_size, r))) for r in range(world_size)]


def imbalance(shares: Sequence[Sequence[int]], costs: Sequence[float]) -> float:
    """Largest share's cost over the mean share's (1.0 = even)."""
    totals = [sum(costs[i] for i in share) for share in shares]
    mean = sum(totals) / max(1, len(totals))
    return max(totals) / mean if mean > 0 else 1.0


def partition_by_cost(costs: Sequence[float], world_size: int, tolerance: float = 0.10) -> list[list[int]]:
    """Which items each rank takes.  Contiguous blocks while their costs are within ``tolerance`` of even ("deal
    round-robin if imbalance > 10 %"); otherwise longest processing time first: items in order of falling cost (ties by
    index), each to the rank with the least work so far (ties to the lowest rank), every share in ascending index order; the
    shares may differ in length (the gather's slots are as wide as the longest).  Deterministic."""
    n = len(costs)
    if world_size < 1:
        raise ValueError("bad world_size")
    blocks = contiguous_shares(n, world_size)
    if world_size == 1 or n == 0 or imbalance(blocks, costs) <= 1.0 + tolerance:
        return blocks
    order = sorted(range(n), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    shares: list[list[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda r: (load[r], r))
        shares[r].append(i)
        load[r] += float(costs[i])
    for share in shares:
        share.sort()
    # (never worse than the blocks it replaces)
    return shares if imbalance(shares, costs) < imbalance(blocks, costs) else blocks


def evaluation_costs(evaluator, circuits: Sequence) -> Optional[list[float]]:
    """Microseconds per evaluation of every circuit on the evaluator's device, or None when it cannot tell (then the shares
    are contiguous blocks).  The same numbers on every rank: a function of the circuit, the operator and the library."""
    tell = getattr(evaluator, "circuit_costs", None)
    if tell is None or os.environ.get("QSV_SHARD_BALANCE", "1") == "0":
        return None
    return [float(c["microseconds"]) for c in tell(list(circuits))]


_SHARES: dict = {}


def population_shares(evaluator, circuits: Sequence, world: int) -> list[list[int]]:
    """:func:`partition_by_cost` of a population, remembered while the same circuit objects come again (an optimiser's next
    iteration): the costs are asked once per list of circuits."""
    n = len(circuits)
    if world == 1:
        return [list(range(n))]
    key = id(evaluator)
    hit = _SHARES.get(key)
    if hit is not None and hit[0]() is evaluator and hit[4] is circuits and len(hit[2]) == n and hit[1][1] == world and (
            n == 0 or (hit[2][0] is circuits[0] and hit[2][n // 2] is circuits[n // 2] and hit[2][-1] is circuits[-1])):
        # (the very list of the last call, a few of its members looked at: walking 64 N ids per step is microseconds per rank
        # that grow with N.  Whatever the list has become in between, the remembered shares are still a partition of its indices:
        # every rank evaluates and returns the right values, at worst less evenly dealt.)
        return hit[3]
    ids = tuple(map(id, circuits))
    if hit is not None and hit[0]() is evaluator and hit[1] == (ids, world):
        _SHARES[key] = hit[:4] + (circuits,)
        return hit[3]
    costs = evaluation_costs(evaluator, circuits)
    shares = contiguous_shares(n, world) if costs is None else partition_by_cost(costs, world)
    try:
        ref = weakref.ref(evaluator, lambda _r, k=key: _SHARES.pop(k, None))
    except TypeError:  # (an evaluator that cannot be weakly referenced: nothing is remembered)
        return shares
    _SHARES[key] = (ref, (ids, world), list(circuits), shares, circuits)  # (holds the circuits: their ids cannot be recycled)
    return shares


def _rows(parameter_values, lo: int, hi: int):
    """Rows [lo, hi) of the population's parameter values: a list of vectors, or -- for a matrix that lives in device memory
    (a torch tensor; ``OperatorCircuitEvaluator.evaluate_circuits`` reads it where it is) -- a view of its rows."""
    if getattr(parameter_values, "is_cuda", False):
        return parameter_values if lo == 0 and hi == len(parameter_values) else parameter_values[lo:hi]
    return list(parameter_values[lo:hi])


def _take(circuits: Sequence, parameter_values, share: Sequence[int]):
    """The share's circuits and parameter values.  A contiguous share of a device matrix is a view of its rows; any other
    share of one is gathered into a new matrix on the device (index_select: one small kernel)."""
    if not share:
        return [], []
    lo, hi = share[0], share[-1] + 1
    if hi - lo == len(share):
        return list(circuits[lo:hi]), _rows(parameter_values, lo, hi)
    picked = [circuits[i] for i in share]
    if getattr(parameter_values, "is_cuda", False):
        import torch

        index = torch.as_tensor(list(share), dtype=torch.int64, device=parameter_values.device)
        return picked, parameter_values.index_select(0, index).contiguous()
    return picked, [parameter_values[i] for i in share]


_GROUP_FACTS: dict = {}


def _group_facts(group):
    """(world size, rank, backend) of a process group, asked once per group object: torch's own accessors walk their registries on
    every call -- tens of microseconds per step on a path whose whole point is a few."""
    import torch.distributed as dist

    g = group if group is not None else dist.group.WORLD
    hit = _GROUP_FACTS.get(id(g))
    if hit is None or hit[0]() is not g:
        hit = (weakref.ref(g), dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group))
        if len(_GROUP_FACTS) > 64:
            _GROUP_FACTS.clear()
        _GROUP_FACTS[id(g)] = hit
    return hit[1], hit[2], hit[3]


def evaluate_population_sharded(evaluator, circuits: Sequence, parameter_values: Sequence, group=None, device=None) -> list[float]:
    """Evaluate this rank's share of the population and all-gather the fitness values.

    ``evaluator`` is any object with ``evaluate_circuits(circuits, parameter_values)``; every rank must pass the
    same full ``circuits`` / ``parameter_values`` lists (or, for evaluators that take it, the population's parameter matrix
    in the rank's own device memory).  Returns all values, ordered by population index, on
    every rank.  Without an initialised process group (or with world size 1) it evaluates everything locally.
    """
    import torch
    import torch.distributed as dist

    n = len(circuits)
    if not (dist.is_available() and dist.is_initialized()):
        return list(evaluator.evaluate_circuits(list(circuits), _rows(parameter_values, 0, n)))
    world, rank, backend = _group_facts(group)
    if world == 1:
        return list(evaluator.evaluate_circuits(list(circuits), _rows(parameter_values, 0, n)))
    shares = population_shares(evaluator, circuits, world)
    if device is None and backend == "nccl" and os.environ.get("QSV_GATHER_CHAIN", "1") != "0":
        # evaluation, collective and copy back as one chain on one stream (no host round trip in between)
        chained = evaluate_block_and_gather(evaluator, circuits, parameter_values, n, world, rank, group,
                                           torch.device("cuda", torch.cuda.current_device()), shares)
        if chained is not None:
            return chained
    if device is None and backend != "nccl":
        # (a CPU group -- rehearsals, tests: the node's shared table serves it too, the host writing the slots)
        # (an evaluator with a GPU of its own under a CPU group -- a gloo rehearsal on a GPU box: the table is registered with
        # that GPU and its kernels store into it, exactly as under RCCL)
        on_gpu = getattr(evaluator, "statevector_device", None) is not None and torch.cuda.is_available()
        table = _node_table(group, world, rank, torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
        if table is not None:
            through_table = evaluate_block_through_node_table(evaluator, circuits, parameter_values, n, world, rank, table, shares,
                                                              torch.cuda.synchronize if on_gpu else None)
            if through_table is not None:
                return through_table
    mine_c, mine_p = _take(circuits, parameter_values, shares[rank])
    local = evaluator.evaluate_circuits(mine_c, mine_p) if mine_c else []
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    owner = id(getattr(evaluator, "statevector_device", None) or evaluator)
    return _gather(local, n, world, rank, group, torch.device(device), owner, shares)


# a receive slot no rank has written yet: a NaN whose payload no arithmetic produces (the evaluator's own sentinel is another)
_SENTINEL = np.frombuffer(np.uint64(0x7FF8C0DEC0DE0001).tobytes(), dtype=np.float64)[0]
_MARKER = _SENTINEL.view(np.uint64)
_POLL_SECONDS = 0.002


class _NodeTable:
    """Two tables of ``world`` slots of ``capacity`` doubles in POSIX shared memory, mapped by every rank of a group whose ranks
    all run on one host and registered with HIP, so that each rank's GPU can store into them (``pointer``: the address the
    device uses), behind one step counter per rank (a cache line each).

    A step: the rank marks ITS slot of table s & 1 with sentinels, starts its evaluation (whose kernels store the values into
    the slot), watches its own slot until no sentinel is left, and only then publishes ``done[rank] = s + 1``; it reads the
    other slots once every rank's counter says s + 1.  (Readers cannot go by the sentinels of somebody else's slot: the
    owner may not have marked it yet, and the values of step s - 2 would pass for this step's.)  Two tables are enough: a
    rank that marks its slot for step s + 2 has seen every counter at s + 2, and a rank publishes that only after it has read
    all of step s.  Created collectively (``create``); rank 0 unlinks the name as soon as everybody has it open."""

    capacity = 4096
    _header = 8  # doubles per rank in front of the tables: the step counter, padded to a cache line

    def __init__(self, mm, world: int, rank: int, device_address: int, host_address: int, registered: bool):
        self.mm, self.world, self.rank = mm, world, rank
        self.device_address, self.host_address, self.registered = device_address, host_address, registered
        self.capacity = type(self).capacity  # (this table's layout, whatever becomes of the class's default)
        self.step = 0
        self.done = np.frombuffer(mm, dtype=np.int64, count=world * self._header)[:: self._header]
        flat = np.frombuffer(mm, dtype=np.float64, count=2 * world * self.capacity, offset=8 * world * self._header)
        self.tables = flat.reshape(2, world, self.capacity)
        self.words = flat.view(np.uint64).reshape(2, world, self.capacity)

    @classmethod
    def bytes_for(cls, world: int) -> int:
        return 8 * (world * cls._header + 2 * world * cls.capacity)

    def pointer(self, table: int, rank: int) -> int:
        return self.device_address + 8 * (self.world * self._header + (table * self.world + rank) * self.capacity)

    def host_pointer(self, table: int, rank: int) -> int:
        return self.host_address + 8 * (self.world * self._header + (table * self.world + rank) * self.capacity)

    @classmethod
    def create(cls, group, world: int, rank: int, device) -> Optional["_NodeTable"]:
        import mmap
        import socket

        import torch
        import torch.distributed as dist

        try:
            boot = open("/proc/sys/kernel/random/boot_id").read().strip()
        except OSError:
            boot = ""
        hosts = [None] * world
        dist.all_gather_object(hosts, (socket.gethostname(), boot), group=group)
        if len(set(hosts)) != 1:
            return None  # (a group that spans nodes: the collective)
        size = cls.bytes_for(world)
        name = [None]
        fd = -1
        if rank == 0:
            path = f"/dev/shm/qsv_fitness_{os.getpid()}_{int.from_bytes(os.urandom(6), 'little'):x}"
            try:
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.ftruncate(fd, size)
                name[0] = path
            except OSError:
                name[0] = None
        dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ok = name[0] is not None
        mm = None
        if ok:
            try:
                if rank != 0:
                    fd = os.open(name[0], os.O_RDWR)
                mm = mmap.mmap(fd, size)
            except OSError:
                ok = False
        if fd >= 0:
            os.close(fd)
        host_address = 0
        device_address = 0
        registered = False
        if ok:
            import ctypes

            host_address = ctypes.addressof(ctypes.c_char.from_buffer(mm))
            device_address = host_address
            if device.type == "cuda":
                try:
                    hip = ctypes.CDLL("libamdhip64.so")
                    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
                    hip.hipHostGetDevicePointer.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_uint]
                    torch.cuda.set_device(device)
                    if hip.hipHostRegister(host_address, size, 3) != 0:  # portable | mapped
                        ok = False
                    else:
                        registered = True
                        seen = ctypes.c_void_p()
                        if hip.hipHostGetDevicePointer(ctypes.byref(seen), host_address, 0) != 0 or not seen.value:
                            ok = False
                        else:
                            device_address = int(seen.value)
                except OSError:
                    ok = False
        # every rank takes the same path: one failure anywhere and nobody uses the table
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok), group=group)
        if rank == 0 and name[0] is not None:
            try:
                os.unlink(name[0])
            except OSError:
                pass
        if not all(flags):
            return None
        table = cls(mm, world, rank, device_address, host_address, registered)
        table.tables[:, rank] = np.nan  # (a fresh file is zeros: the counters start at 0)
        dist.barrier(group=group)
        return table


_NODE_TABLES: dict = {}


def _node_table(group, world: int, rank: int, device):
    """The group's table (created on first use, collectively), or None: switched off, a group over several nodes, no shared
    memory, or memory the device cannot be given."""
    if os.environ.get("QSV_GATHER_NODE", "1") == "0":
        return None
    key = (id(group) if group is not None else 0, world, str(device))
    hit = _NODE_TABLES.get(key)
    if hit is None:
        made = _NodeTable.create(group, world, rank, device)
        hit = (made, weakref.ref(group) if group is not None else None)
        _NODE_TABLES[key] = hit
    table, ref = hit
    if ref is not None and ref() is not group:  # (an id() recycled by another group)
        _NODE_TABLES.pop(key, None)
        return _node_table(group, world, rank, device)
    return table


_NODE_WAIT_SECONDS = 120.0
_WAITER: list = []


def _table_waiter():
    """``qsv_fitness_table_wait`` of the library (a spin in C without the interpreter lock: own slot, publish, the other ranks'
    counters), or None where the library is not there (it is never built for this: a CPU rank with a foreign evaluator)."""
    if not _WAITER:
        fn = None
        if os.environ.get("QSV_GATHER_NODE_SPIN", "1") != "0":
            try:
                from . import _lib

                fn = getattr(_lib.load(build_if_missing=False), "qsv_fitness_table_wait", None)
            except Exception:  # (no library, or one from before this entry point)
                fn = None
        _WAITER.append(fn)
    return _WAITER[0]


def evaluate_block_through_node_table(evaluator, circuits: Sequence, parameter_values: Sequence, n: int, world: int, rank: int,
                                      table: "_NodeTable", shares: list, synchronize=None) -> Optional[list[float]]:
    """One step through the node's shared table: this rank's share evaluated with the kernels' results going straight into
    the rank's slot (an evaluator that cannot do that evaluates the ordinary way and the host writes the slot), then every
    rank's values read off the table.  No collective call.  ``synchronize``: waits for this rank's own device work (called when
    the table has not filled after a couple of milliseconds: a long step needs no busy host).  None: the shares do not fit."""
    width = max(len(share) for share in shares)
    if width > table.capacity:
        return None
    mine_c, mine_p = _take(circuits, parameter_values, shares[rank])
    count = len(mine_c)
    which = table.step & 1
    table.step += 1
    step = table.step
    slot = table.tables[which, rank]
    slot[:count] = _SENTINEL
    if count < width:
        slot[count:width] = np.nan
    to_device = getattr(evaluator, "evaluate_circuits_to_device", None)
    if not table.registered and getattr(evaluator, "statevector_device", None) is not None:
        to_device = None  # (memory the evaluator's GPU has not been given: the host writes the slot)
    launched = bool(count) and to_device is not None and to_device(mine_c, mine_p, table.pointer(which, rank))
    if count and not launched:
        slot[:count] = evaluator.evaluate_circuits(mine_c, mine_p)
    seen = getattr(getattr(evaluator, "statevector_device", None), "results_seen", None) if launched else None
    waiter = _table_waiter()
    if waiter is not None and waiter(table.host_pointer(which, rank), count, table.host_address, table._header, world, rank, step,
                                     int(_POLL_SECONDS * 1e6)) == 0:
        if seen is not None:
            seen()  # (every value of this rank's batch has arrived: the library need not wait for it before its next batch)
        return _unpack(table.tables[which, :, :width], n, world, width, shares)
    # (without the library -- a CPU rank whose evaluator is not ours -- or past the spinning budget: the same in NumPy, with
    # this rank's own device work waited for once and a deadline)
    mine = table.words[which, rank, :count]
    done = table.done
    t0 = None
    waited = False
    # this rank's own values (from its own kernels), then every rank's counter
    for own in (True, False):
        while (mine == _MARKER).any() if own else (done < step).any():
            if t0 is None:
                t0 = time.perf_counter()
                continue
            elapsed = time.perf_counter() - t0
            if elapsed > _POLL_SECONDS and not waited:
                waited = True  # (a long step needs no busy host: wait for this rank's own device work once)
                if synchronize is not None:
                    synchronize()
            elif elapsed > _NODE_WAIT_SECONDS:
                raise RuntimeError("queasars_amd.distributed: " + ("this rank's" if own else "another rank's") +
                                   f" fitness values did not reach the node's shared table within {_NODE_WAIT_SECONDS:.0f} s")
            elif waited:
                time.sleep(0)
        if own:
            done[rank] = step
            if seen is not None:
                seen()
    return _unpack(table.tables[which, :, :width], n, world, width, shares)


def evaluate_block_and_gather(evaluator, circuits: Sequence, parameter_values: Sequence, n: int, world: int, rank: int, group, device,
                              shares: Optional[list] = None):
    """This rank's share and the all-gather as ONE chain on one HIP stream: the
    evaluator leaves its values in the collective's send buffer (``evaluate_circuits_to_device``: device memory, no wait),
    the all-gather follows on the same stream.  Measured on one MI355X (nccl group of one rank, ``scripts/gatherstep.py``):
    round 2, the step with the gather cost 138 us the staged way (evaluate, wait, stage, copy to the device, gather, copy
    back, wait) against 91 us for the evaluation alone; round 3, chained with a copy back and a stream synchronisation at its
    end, 27 + 4 us over the evaluation; round 4: the collective RECEIVES INTO HOST MEMORY the device can address (a pinned
    buffer seen as a device tensor) and the host reads the end of the step off the slots themselves -- every slot starts as
    a sentinel no arithmetic produces, the values are there when no sentinel is left -- as ``qsv_eval_end`` does with its
    result buffer: no copy back, no synchronisation call (``QSV_GATHER_HOST=0``: the copy back).  Returns None when the
    evaluator cannot leave its values on the device (the caller then takes the staged way).

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
    if shares is None:
        shares = contiguous_shares(n, world)
    stream = state["stream"]
    if not state.get("no_node_table"):
        # One node: no collective -- the kernels' results go straight into this rank's slot of the shared table.  Nothing of the
        # collective's apparatus is touched on this path (buffers, the current-stream switch: the evaluator launches on its own
        # stream anyway); the caller's stream is waited for only if it has work queued.
        node_table = _node_table(group, world, rank, device)
        if node_table is not None:
            # (the caller's current stream by its raw handle first: building the Stream object costs more than the table look-ups)
            raw = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())
            if raw != stream.cuda_stream:
                caller = state.get("default_stream") if raw == 0 else None
                if caller is None:
                    caller = torch.cuda.current_stream(device)
                    if raw == 0:
                        state["default_stream"] = caller
                if not caller.query():
                    stream.wait_stream(caller)
            values = evaluate_block_through_node_table(evaluator, circuits, parameter_values, n, world, rank, node_table, shares,
                                                       stream.synchronize)
            if values is not None:
                if state.get("node_verified") is node_table:
                    return values
                mine_c, mine_p = _take(circuits, parameter_values, shares[rank])
                checked = _verified_or_staged(evaluator, values, mine_c, mine_p, n, world, rank, group, device, state, shares, "node")
                if not state.get("no_node_table"):
                    state["node_verified"] = node_table
                return checked
    mine_c, mine_p = _take(circuits, parameter_values, shares[rank])
    width = max(len(share) for share in shares)
    host_receive = os.environ.get("QSV_GATHER_HOST", "1") != "0" and not state.get("no_host_receive")
    _, send, recv, recv_host = _buffers(world, width, device, group, state["key"])
    mapped = _mapped_receive(world, width, device, group, state["key"]) if host_receive else None
    if host_receive and mapped is None:
        state["no_host_receive"] = True
        host_receive = False
    caller = torch.cuda.current_stream(device)
    if caller != stream:
        stream.wait_stream(caller)  # (the application's earlier work on its own stream comes first)
        torch.cuda.set_stream(stream)
    try:
        if len(mine_c) < width:
            send.fill_(float("nan"))
        recv_values = _collective_step(to_device, mine_c, mine_p, send, recv, recv_host, mapped, host_receive, state, group, stream)
        if recv_values is None:
            return None
        values = _unpack(recv_values, n, world, width, shares)
        way = "host" if host_receive and not state.get("no_host_receive") else "device"
    finally:
        if caller != stream:
            torch.cuda.set_stream(caller)
    return _verified_or_staged(evaluator, values, mine_c, mine_p, n, world, rank, group, device, state, shares, way)


def _collective_step(to_device, mine_c, mine_p, send, recv, recv_host, mapped, host_receive, state, group, stream):
    """The RCCL way of :func:`evaluate_block_and_gather`: the evaluation's values into the send buffer, the all-gather behind it
    on the same stream -- into host-mapped memory read off its sentinels, or into device memory and copied back.  The
    gathered slots, or None where the evaluator cannot leave its values on the device."""
    import torch.distributed as dist

    if host_receive:
        mapped[1].fill(_SENTINEL)
    if mine_c and not to_device(mine_c, mine_p, send.data_ptr()):
        return None
    if host_receive:
        try:
            dist.all_gather_into_tensor(mapped[0], send, group=group)
        except Exception as exc:  # (a collective library that refuses host-mapped memory refuses it on every rank alike)
            import warnings

            warnings.warn(f"queasars_amd.distributed: all-gather into host-mapped memory refused ({exc}); receiving on the device",
                          RuntimeWarning)
            state["no_host_receive"] = True
            host_receive = False
    if host_receive:
        table = mapped[1]
        deadline = time.perf_counter() + _POLL_SECONDS
        flat = table.view(np.uint64)
        marker = np.float64(_SENTINEL).view(np.uint64)
        while (flat == marker).any():
            if time.perf_counter() > deadline:  # (a long step: wait on the stream as before)
                stream.synchronize()
                break
        return table
    dist.all_gather_into_tensor(recv, send, group=group)
    recv_host.copy_(recv, non_blocking=True)
    stream.synchronize()
    return recv_host.numpy()


def _verified_or_staged(evaluator, values, mine_c, mine_p, n, world, rank, group, device, state, shares, way: str):
    """The first chained step of an evaluator on a group, run the staged way as well and compared (every rank agrees on the
    verdict); a way that disagrees is switched off for that evaluator, with a warning, and the staged values are returned."""
    import torch
    import torch.distributed as dist

    host_receive = way == "host"
    verified = state["verified"]  # (group, way) pairs this evaluator's chained step has been checked on
    known = (any(ref() is group and w == way for ref, w in verified) if group is not None
             else way in state.setdefault("verified_default", set()))
    if not known:
        local = evaluator.evaluate_circuits(mine_c, mine_p) if mine_c else []
        staged = _gather(local, n, world, rank, group, device, state["key"], shares)
        same = len(staged) == len(values) and all(a == b for a, b in zip(staged, values))
        # every rank must take the same path from now on: agree on the verdict (a collective itself, staged way)
        flag = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag.item()) != 1.0:
            import warnings

            if way == "node":
                warnings.warn("queasars_amd.distributed: the step through the node's shared table disagreed with the staged step; "
                              "using the collective", RuntimeWarning)
                state["no_node_table"] = True
                return staged
            if host_receive:  # (first suspect: the collective writing into host memory; the device receive buffer gets its turn)
                warnings.warn("queasars_amd.distributed: the all-gather into host-mapped memory disagreed with the staged step; "
                              "receiving on the device", RuntimeWarning)
                state["no_host_receive"] = True
                return staged
            warnings.warn("queasars_amd.distributed: the chained evaluate + all-gather step disagreed with the staged one; "
                          "using the staged path", RuntimeWarning)
            state["disabled"] = True
            return staged
        if group is not None:
            # (by the group object itself, weakly: an id() can be recycled by another group)
            verified.append((weakref.ref(group), way))
            verified[:] = [(ref, w) for ref, w in verified if ref() is not None]
        else:
            state["verified_default"].add(way)
    return values


def _unpack(recv_values, n: int, world: int, width: int, shares: Optional[list] = None) -> list[float]:
    """The gathered slots (one of ``width`` values per rank, the unused tail NaN) as one list ordered by population index."""
    table = np.asarray(recv_values).reshape(world, width)
    if shares is None:
        shares = contiguous_shares(n, world)
    if n == world * width:
        even = _EVEN_BLOCKS.get(id(shares))
        if even is None or even[0] is not shares:  # (asked once per shares object: N lists of 64 N indices compared per step otherwise)
            even = (shares, all(share == list(range(r * width, (r + 1) * width)) for r, share in enumerate(shares)))
            if len(_EVEN_BLOCKS) > 64:
                _EVEN_BLOCKS.clear()
            _EVEN_BLOCKS[id(shares)] = even
        if even[1]:
            return table.ravel().tolist()
    out = [0.0] * n
    for r, share in enumerate(shares):
        row = table[r, : len(share)].tolist()
        for i, v in zip(share, row):
            out[i] = v
    return out


_EVEN_BLOCKS: dict = {}
_CHAIN_STATE: dict = {}


def _chain_state(evaluator, device):
    """Per evaluator device: the HIP stream it launches on as a torch stream (created once and handed to the library: its
    kernels, the collective and the copy back are then ordered by the stream alone), which groups the chained step has
    been verified on, and the key of its buffers.  The entry goes when the device does."""
    import torch

    dev = getattr(evaluator, "statevector_device", None)
    if dev is None:
        return None
    hit = _CHAIN_STATE.get(id(dev))
    if hit is None or hit["ref"]() is not dev:
        key = id(dev)

        def gone(_ref, key=key, tables=(_CHAIN_STATE, _BUFFERS, _MAPPED)):  # (bound now: at interpreter exit the globals are gone)
            tables[0].pop(key, None)
            for table in tables[1:]:
                for k in [k for k in table if k[-1] == key]:
                    table.pop(k, None)

        stream = torch.cuda.Stream(device=device)
        dev.set_stream(stream.cuda_stream)
        hit = {"ref": weakref.ref(dev, gone), "stream": stream, "verified": [], "disabled": False, "key": key}
        _CHAIN_STATE[key] = hit
    return hit


def _gather(local: Sequence[float], n: int, world: int, rank: int, group, device, owner=None, shares: Optional[list] = None) -> list[float]:
    """All ranks' shares of fitness values, ordered by population index (``local`` is this rank's share)."""
    import torch
    import torch.distributed as dist

    # every rank contributes a fixed-size slot so one all_gather_into_tensor suffices: as wide as the longest share
    width = -(-n // world) if shares is None else max(len(share) for share in shares)
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
    return _unpack(recv_host.numpy(), n, world, width, shares)


_BUFFERS: dict = {}
_MAPPED: dict = {}


def _buffer_key(world: int, width: int, device, group, owner):
    import threading

    return (world, width, str(device), id(group) if group is not None else 0, threading.get_ident(), owner)


def _buffers(world: int, width: int, device, group=None, owner=None):
    """(pinned send staging, device send, device receive, pinned receive staging) for one (world, width, device, group,
    calling thread, owner): two evaluators, groups or threads never share a send / receive buffer (``owner`` = the
    evaluator device of a chained step; its entries are dropped when it dies)."""
    import torch

    key = _buffer_key(world, width, device, group, owner)
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


class _DeviceView:
    """Pinned host memory presented through ``__cuda_array_interface__``: on ROCm a pinned allocation has ONE address, valid on
    the host and on every device, so ``torch.as_tensor(view, device=...)`` is a device tensor whose bytes live in host memory."""

    def __init__(self, pinned):
        self._keep = pinned
        self.__cuda_array_interface__ = {"shape": tuple(pinned.shape), "typestr": "<f8", "data": (pinned.data_ptr(), False),
                                         "version": 2, "strides": None}


def _mapped_receive(world: int, width: int, device, group=None, owner=None):
    """(device tensor over pinned host memory, the same memory as a NumPy array) or None where torch will not make one."""
    import torch

    key = _buffer_key(world, width, device, group, owner)
    hit = _MAPPED.get(key)
    if hit is None:
        try:
            pinned = torch.empty(world * width, dtype=torch.float64, pin_memory=True)
            as_device = torch.as_tensor(_DeviceView(pinned), device=device)
            if as_device.data_ptr() != pinned.data_ptr() or not as_device.is_cuda:
                return None
        except Exception:  # (no unified addressing, or torch refuses the interface: the device receive buffer is used)
            return None
        hit = (as_device, pinned.numpy(), pinned)
        _MAPPED[key] = hit
    return hit


# ---- layer searches sharded by individual ---------------------------------------------------------------------------------


def shard_searches(evaluator, jobs: Sequence, group=None) -> Optional[list[int]]:
    """Which of a layer search's (circuit, run) jobs THIS rank runs (reference: every individual's whole optimiser run is one task
    on the pool, mutation.py:194-235): the jobs dealt by what their circuits cost (:func:`partition_by_cost`), or None without
    a process group of several ranks."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    costs = evaluation_costs(evaluator, [circuit for circuit, _ in jobs])
    shares = contiguous_shares(len(jobs), world) if costs is None else partition_by_cost(costs, world)
    return shares[rank]


def gather_search_results(jobs: Sequence, mine: Sequence[int], group=None) -> None:
    """After every rank has advanced ITS runs to completion: ONE all-gather per search hands every rank every run's final
    iterate, iteration count and evaluation count (mutation.py:206-218: the futures' results), so that the evolution goes on
    identically everywhere.  Rows of max(len(x)) + 3 doubles: job index, iterations, evaluations, the iterate."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = len(jobs)
    width = max((run.x.size for _, run in jobs), default=0) + 3
    rows = -(-n // world)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    send = np.full((rows, width), np.nan)
    for slot, j in enumerate(mine):
        run = jobs[j][1]
        send[slot, 0], send[slot, 1], send[slot, 2] = j, run.iteration, run.nfev
        send[slot, 3 : 3 + run.x.size] = run.x
    send_t = torch.from_numpy(send).to(device)
    recv_t = torch.empty((world * rows, width), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(recv_t, send_t, group=group)
    table = recv_t.cpu().numpy()
    seen = set()
    for row in table:
        if np.isnan(row[0]):
            continue
        j = int(row[0])
        run = jobs[j][1]
        run.x = row[3 : 3 + run.x.size].copy()
        run.iteration, run.nfev, run.done = int(row[1]), int(row[2]), True
        seen.add(j)
    if len(seen) != n:
        raise RuntimeError(f"sharded search: results of {n - len(seen)} of {n} runs are missing")
