"""JSSP -> diagonal Hamiltonian in the domain-wall encoding (the operator of BASELINE config 4).

Restates, as a direct (z_mask, coefficient) generator with no Qiskit objects, what the reference builds with
``SparsePauliOp`` algebra:

* problem datatypes: queasars/job_shop_scheduling/problem_instances.py:10-120 (Machine, Operation, Job, instance),
  validity / makespan of a schedule :289-427;
* domain-wall variables: queasars/utility/domain_wall_variables.py:43-143 (``_z_dash_term``, ``viability_term``,
  ``value_term``), decoding :145-172;
* the encoder: queasars/job_shop_scheduling/domain_wall_hamiltonian_encoder.py -- qubit assignment :146-187,
  Hamiltonian assembly :189-230, overlap :232-276, precedence :278-320, makespan term :322-347, early-start
  term :349-371, bitstring decoding :107-144.

Every term is a product of I and Z factors, so the operator is a polynomial in commuting Z's; it is kept as a
``{z_mask: coefficient}`` map (multiplying two terms XORs their masks).  The result is a diagonal
:class:`~queasars_amd.ir.PauliOperator`, which takes the evaluator's diagonal fast path.
"""

from __future__ import annotations

from dataclasses import dataclass
from itertools import combinations
from typing import Optional

from queasars_amd.ir import PauliOperator


class JobShopSchedulingProblemException(Exception):
    pass


@dataclass(frozen=True)
class Machine:
    name: str

    def __post_init__(self):
        if self.name == "":
            raise JobShopSchedulingProblemException("The name of a Machine cannot be an empty string!")

    def __repr__(self) -> str:  # (the texts a notebook prints are the reference's, line for line: tests/golden/jssp_reference.json)
        return self.name


@dataclass(frozen=True)
class Operation:
    name: str
    job_name: str
    machine: Machine
    processing_duration: int

    def __post_init__(self):
        if self.name == "" or self.job_name == "":
            raise JobShopSchedulingProblemException("Operation and job names cannot be empty strings!")
        if self.processing_duration <= 0:
            raise JobShopSchedulingProblemException(
                f"The processing_duration of an Operation must at least be one, but it was {self.processing_duration}"
            )

    @property
    def identifier(self) -> str:
        return self.job_name + "_" + self.name

    def __repr__(self) -> str:
        return f"{self.identifier}({self.machine.name}, {self.processing_duration})"


@dataclass(frozen=True)
class Job:
    name: str
    operations: tuple[Operation, ...]

    def __post_init__(self):
        if self.name == "":
            raise JobShopSchedulingProblemException("The name of a Job cannot be an empty string!")
        if len(self.operations) == 0:
            raise JobShopSchedulingProblemException("A job must contain at least 1 operation!")
        if len({op.identifier for op in self.operations}) != len(self.operations):
            raise JobShopSchedulingProblemException("The identifiers of all operations within a job must be unique!")
        machines = [op.machine for op in self.operations]
        if len(set(machines)) != len(machines):
            raise JobShopSchedulingProblemException("A job may visit every machine at most once!")
        if any(op.job_name != self.name for op in self.operations):
            raise JobShopSchedulingProblemException("Every operation must carry the name of its job!")

    def __repr__(self) -> str:
        return f"{self.name}:\n" + "".join(f"  {operation!r}\n" for operation in self.operations)

    def is_consistent_with_machines(self, machines: tuple[Machine, ...]) -> bool:
        """Does every operation of the job run on one of ``machines``?  (reference: problem_instances.py:91-103)"""
        return all(op.machine in machines for op in self.operations)


@dataclass(frozen=True)
class JobShopSchedulingProblemInstance:
    name: str
    machines: tuple[Machine, ...]
    jobs: tuple[Job, ...]

    def __post_init__(self):
        if self.name == "":
            raise JobShopSchedulingProblemException("The name of a problem instance cannot be an empty string!")
        if len({m.name for m in self.machines}) != len(self.machines):
            raise JobShopSchedulingProblemException("Machine names must be unique!")
        if len({j.name for j in self.jobs}) != len(self.jobs):
            raise JobShopSchedulingProblemException("Job names must be unique!")
        for job in self.jobs:
            if not job.is_consistent_with_machines(self.machines):
                raise JobShopSchedulingProblemException("A job uses a machine the instance does not have!")

    def __repr__(self) -> str:
        lines = [self.name, "  Machines:"] + [f"    {machine!r}" for machine in self.machines] + ["  Jobs:"]
        for job in self.jobs:
            lines += ["    " + line for line in repr(job).splitlines()]
        return "\n".join(lines) + "\n"


@dataclass(frozen=True)
class PotentiallyScheduledOperation:
    """An operation of a result's schedule: with a start time (:class:`ScheduledOperation`) or without one
    (:class:`UnscheduledOperation`: the qubits of its variable hold no valid domain-wall state).  Reference:
    problem_instances.py:204-266."""

    operation: Operation

    @property
    def is_scheduled(self) -> bool:
        raise NotImplementedError


@dataclass(frozen=True)
class UnscheduledOperation(PotentiallyScheduledOperation):
    @property
    def is_scheduled(self) -> bool:
        return False

    def __repr__(self) -> str:
        return f"{self.operation!r} was not scheduled."


@dataclass(frozen=True)
class ScheduledOperation(PotentiallyScheduledOperation):
    start_time: int

    @property
    def is_scheduled(self) -> bool:
        return True

    @property
    def end_time(self) -> int:
        return self.start_time + self.operation.processing_duration

    def __repr__(self) -> str:
        return f"{self.operation!r} starts at: {self.start_time} and ends at: {self.end_time}"


class JobShopSchedulingResult:
    """A (possibly incomplete, possibly infeasible) schedule of a problem instance: per job one entry per operation, in the
    job's order (reference: problem_instances.py:289-427).  Valid = every operation scheduled, the operations of a job one
    after the other, no two operations on a machine at the same time; the makespan of a valid schedule is its latest end."""

    def __init__(self, problem_instance: JobShopSchedulingProblemInstance,
                 schedule: dict[Job, tuple[PotentiallyScheduledOperation, ...]]):
        if set(schedule) != set(problem_instance.jobs):
            raise JobShopSchedulingProblemException("The schedule must hold exactly the jobs of the problem instance!")
        for job, entries in schedule.items():
            if tuple(entry.operation for entry in entries) != job.operations:
                raise JobShopSchedulingProblemException("A job's schedule must hold the job's operations, in their order!")
        self._problem_instance = problem_instance
        self._schedule = {job: tuple(entries) for job, entries in schedule.items()}
        self._is_valid = self._check()

    @classmethod
    def from_start_times(cls, problem_instance: JobShopSchedulingProblemInstance,
                         start_times: dict[Operation, Optional[int]]) -> "JobShopSchedulingResult":
        """From a start time per operation (``None`` = not scheduled)."""
        return cls(problem_instance, {
            job: tuple(UnscheduledOperation(op) if start_times[op] is None else ScheduledOperation(op, start_times[op])
                       for op in job.operations)
            for job in problem_instance.jobs})

    @property
    def problem_instance(self) -> JobShopSchedulingProblemInstance:
        return self._problem_instance

    @property
    def schedule(self) -> dict[Job, tuple[PotentiallyScheduledOperation, ...]]:
        return self._schedule

    @property
    def valid_schedule(self) -> dict[Job, tuple[ScheduledOperation, ...]]:
        if not self._is_valid:
            raise JobShopSchedulingProblemException("The schedule is not valid!")
        return self._schedule  # (every entry of a valid schedule is a ScheduledOperation)

    @property
    def start_times(self) -> dict[Operation, Optional[int]]:
        """Start time per operation, ``None`` where it has none."""
        return {entry.operation: (entry.start_time if entry.is_scheduled else None)
                for entries in self._schedule.values() for entry in entries}

    def _check(self) -> bool:
        per_machine: dict[Machine, list[tuple[int, int]]] = {m: [] for m in self._problem_instance.machines}
        for job in self._problem_instance.jobs:
            previous_end = None
            for entry in self._schedule[job]:
                if not entry.is_scheduled:
                    return False
                if previous_end is not None and entry.start_time < previous_end:
                    return False
                previous_end = entry.end_time
                per_machine[entry.operation.machine].append((entry.start_time, entry.end_time))
        for spans in per_machine.values():
            spans.sort()
            if any(b[0] < a[1] for a, b in zip(spans, spans[1:])):
                return False
        return True

    @property
    def is_valid(self) -> bool:
        return self._is_valid

    @property
    def makespan(self) -> Optional[int]:
        if not self._is_valid:
            return None
        return max(entries[-1].end_time for entries in self._schedule.values())

    def __eq__(self, other) -> bool:
        return (isinstance(other, JobShopSchedulingResult) and self._problem_instance == other._problem_instance
                and self._schedule == other._schedule)

    def __hash__(self) -> int:
        return hash((self._problem_instance, tuple(sorted((job.name, entries) for job, entries in self._schedule.items()))))

    def __repr__(self) -> str:
        lines = [f"{self._problem_instance.name} solution with makespan {self.makespan}"]
        for job in self._problem_instance.jobs:
            lines.append(f"  {job.name}:")
            lines += [f"    {entry!r}" for entry in self._schedule[job]]
        return "\n".join(lines) + "\n"


class _Poly(dict):
    """Polynomial in commuting Pauli-Z's: {z_mask: coefficient}; mask 0 is the identity."""

    def add(self, other: "_Poly", scale: float = 1.0) -> "_Poly":
        for mask, c in other.items():
            self[mask] = self.get(mask, 0.0) + scale * c
        return self

    def times(self, other: "_Poly") -> "_Poly":
        out = _Poly()
        for m1, c1 in self.items():
            for m2, c2 in other.items():
                out[m1 ^ m2] = out.get(m1 ^ m2, 0.0) + c1 * c2
        return out

    def scaled(self, scale: float) -> "_Poly":
        return _Poly({m: scale * c for m, c in self.items()})


class DomainWallVariable:
    """Choice between len(values) values on len(values) - 1 qubits starting at ``qubit_start_index``: value i is
    chosen when the first i qubits are 1 and the rest 0."""

    def __init__(self, qubit_start_index: int, values: tuple):
        if len(values) < 1:
            raise ValueError("The domain wall variable must at least have one value!")
        if len(set(values)) != len(values):
            raise ValueError("All values of a domain wall variable must be unique!")
        self._start = qubit_start_index
        self.values = tuple(values)
        self.n_qubits = len(values) - 1

    def _z_dash(self, i: int) -> _Poly:
        if i < -1 or i > self.n_qubits:
            raise ValueError("The index is out of the bounds of the domain wall variable!")
        if i == -1:
            return _Poly({0: -1.0})  # virtual qubit before the variable: fixed to 1
        if i == self.n_qubits:
            return _Poly({0: 1.0})   # virtual qubit after the variable: fixed to 0
        return _Poly({1 << (self._start + i): 1.0})

    def viability_term(self) -> _Poly:
        """0 on valid states, (number of domain walls - 1) otherwise."""
        out = _Poly()
        if self.n_qubits == 0:
            return out
        for i in range(-1, self.n_qubits):
            out.add(_Poly({0: 0.5}))
            out.add(self._z_dash(i).times(self._z_dash(i + 1)), -0.5)
        out.add(_Poly({0: -1.0}))
        return out

    def value_term(self, value) -> _Poly:
        """1 on states in which the variable holds ``value``, 0 on the other valid states."""
        if value not in self.values:
            raise ValueError("The domain wall variable can never assume this value!")
        if self.n_qubits == 0:
            return _Poly({0: 1.0})
        i = self.values.index(value)
        return _Poly().add(self._z_dash(i), 0.5).add(self._z_dash(i - 1), -0.5)

    def value_from_bits(self, bits: list[int]):
        mine = bits[self._start : self._start + self.n_qubits]
        wall = self.n_qubits
        for i, b in enumerate(mine):
            if b == 0:
                wall = i
                break
        if sum(mine[wall:]) != 0:
            return None
        return self.values[wall]


class JSSPDomainWallHamiltonianEncoder:
    """Time-indexed JSSP model with domain-wall start-time variables, as a diagonal Hamiltonian."""

    def __init__(
        self,
        jssp_instance: JobShopSchedulingProblemInstance,
        makespan_limit: int,
        encoding_penalty: float = 300,
        overlap_constraint_penalty: float = 100,
        precedence_constraint_penalty: float = 100,
        max_opt_value: float = 100,
        opt_all_operations_share: float = 0,
    ):
        self.jssp_instance = jssp_instance
        self.makespan_limit = makespan_limit
        self._penalties = (encoding_penalty, overlap_constraint_penalty, precedence_constraint_penalty)
        self._max_opt_value = max_opt_value
        self._share = opt_all_operations_share
        self._variables: dict[Operation, DomainWallVariable] = {}
        self._machine_operations: dict[Machine, list[Operation]] = {}
        self._constraint_counts: dict[tuple[Operation, int], int] = {}
        self._n_qubits = 0
        self._hamiltonian: Optional[PauliOperator] = None
        for job in jssp_instance.jobs:
            start_offset, end_offset = 0, sum(op.processing_duration for op in job.operations)
            if end_offset > makespan_limit:
                raise ValueError(
                    f"There is no feasible solution for the given makespan_limit {makespan_limit}!\\n"
                    + f"This is due to the length of all operations in job {job.name} which\\n"
                    + f"is {end_offset} and is longer than the makespan_limit!"
                )
            for op in job.operations:
                self._machine_operations.setdefault(op.machine, []).append(op)
                n_start_times = makespan_limit - (start_offset + end_offset) + 1
                var = DomainWallVariable(self._n_qubits, tuple(range(start_offset, start_offset + n_start_times)))
                self._variables[op] = var
                for s in var.values:
                    self._constraint_counts[(op, s)] = 0
                self._n_qubits += var.n_qubits
                start_offset += op.processing_duration
                end_offset -= op.processing_duration

    @property
    def n_qubits(self) -> int:
        return self._n_qubits

    def _pair_penalty(self, op1: Operation, op2: Operation, violated) -> _Poly:
        v1, v2 = self._variables[op1], self._variables[op2]
        out = _Poly()
        for s1 in v1.values:
            for s2 in v2.values:
                if violated(s1, s2):
                    self._constraint_counts[(op1, s1)] += 1
                    self._constraint_counts[(op2, s2)] += 1
                    out.add(v1.value_term(s1).times(v2.value_term(s2)))
        return out

    def _precedence_term(self, op1: Operation, op2: Operation) -> _Poly:
        v1, v2 = self._variables[op1], self._variables[op2]
        if v1.values[-1] + op1.processing_duration <= v2.values[0]:
            return _Poly()
        return self._pair_penalty(op1, op2, lambda s1, s2: not s1 + op1.processing_duration <= s2)

    def _overlap_term(self, op1: Operation, op2: Operation) -> _Poly:
        v1, v2 = self._variables[op1], self._variables[op2]
        if v1.values[-1] + op1.processing_duration <= v2.values[0]:
            return _Poly()
        if v2.values[-1] + op2.processing_duration <= v1.values[0]:
            return _Poly()
        return self._pair_penalty(
            op1, op2, lambda s1, s2: s1 < s2 + op2.processing_duration and s2 < s1 + op1.processing_duration
        )

    def _makespan_term(self) -> _Poly:
        n_jobs = len(self.jssp_instance.jobs)
        norm = n_jobs * (n_jobs + 1) ** self.makespan_limit
        out = _Poly()
        for job in self.jssp_instance.jobs:
            last = job.operations[-1]
            var = self._variables[last]
            for s in var.values:
                out.add(var.value_term(s), (1 / norm) * (n_jobs + 1) ** (s + last.processing_duration))
        return out

    def _early_start_term(self) -> _Poly:
        norm = sum(len(v.values) - 1 for v in self._variables.values())
        out = _Poly()
        for var in self._variables.values():
            for i, value in enumerate(var.values):
                if i:
                    out.add(var.value_term(value), (1 / norm) * i)
        return out

    def get_problem_hamiltonian(self) -> PauliOperator:
        if self._hamiltonian is not None:
            return self._hamiltonian
        encoding_penalty, overlap_penalty, precedence_penalty = self._penalties
        precedence = _Poly()
        for job in self.jssp_instance.jobs:
            for a, b in zip(job.operations, job.operations[1:]):
                precedence.add(self._precedence_term(a, b))
        overlap = _Poly()
        for operations in self._machine_operations.values():
            for a, b in combinations(operations, 2):
                overlap.add(self._overlap_term(a, b))
        viability = _Poly()
        for job in self.jssp_instance.jobs:
            for op in job.operations:
                var = self._variables[op]
                worst = max([self._constraint_counts[(op, s)] for s in var.values] + [0])
                viability.add(var.viability_term(), worst + 1)
        total = _Poly()
        total.add(precedence, precedence_penalty)
        total.add(overlap, overlap_penalty)
        total.add(viability, encoding_penalty)
        total.add(self._makespan_term(), self._max_opt_value * (1 - self._share))
        if self._early_start_norm():
            total.add(self._early_start_term(), self._max_opt_value * self._share)
        masks = sorted(total)
        n = max(self._n_qubits, 1)
        labels = ["".join("Z" if (m >> (n - 1 - pos)) & 1 else "I" for pos in range(n)) for m in masks]
        self._hamiltonian = PauliOperator(labels, [total[m] for m in masks])
        return self._hamiltonian

    def _early_start_norm(self) -> int:
        return sum(len(v.values) - 1 for v in self._variables.values())

    def translate_result_bitstring(self, bitstring: str) -> JobShopSchedulingResult:
        """Measured bitstring (qiskit order: last character = qubit 0) -> schedule."""
        if len(bitstring) != self._n_qubits:
            raise ValueError("The bitstring length does not match the problem size!")
        if set(bitstring) - {"0", "1"}:
            raise ValueError("The bitstring may not contain any value apart from 1 or 0!")
        bits = [int(ch) for ch in bitstring[::-1]]
        return JobShopSchedulingResult.from_start_times(
            self.jssp_instance, {op: var.value_from_bits(bits) for op, var in self._variables.items()}
        )

    def bitstring_of(self, start_times: dict[Operation, int]) -> str:
        """Inverse of :meth:`translate_result_bitstring` for a complete assignment of start times."""
        bits = [0] * self._n_qubits
        for op, var in self._variables.items():
            wall = var.values.index(start_times[op])
            for i in range(wall):
                bits[var._start + i] = 1
        return "".join(str(b) for b in bits[::-1])


# ---- random instances (reference: queasars/job_shop_scheduling/random_problem_instances.py:50-101) ------------------------


def random_job_shop_scheduling_instance(instance_name: str, n_jobs: int, n_machines: int, relative_op_amount, op_duration,
                                        random_seed: Optional[int] = None) -> JobShopSchedulingProblemInstance:
    """A random instance with machines ``m0 ..`` and jobs ``job0 ..`` whose operations ``op0 ..`` visit a random selection of
    the machines in random order.  ``relative_op_amount`` (operations per job as a share of the machines) and ``op_duration``
    are either plain values or distributions ``{value: probability}``.  The generator is consumed exactly as the reference
    consumes it -- per job: the share (if it is a distribution), a sample of the machines, a shuffle, then one duration per
    operation (if durations are a distribution) -- so a seed names the same instance on both sides (held to vectors the
    reference produced: tests/golden/jssp_reference.json)."""
    from math import isclose
    from random import Random

    rng = Random(random_seed)

    def draw(value_or_distribution):
        if not isinstance(value_or_distribution, dict):
            return value_or_distribution
        if not isclose(sum(value_or_distribution.values()), 1, abs_tol=0.001):
            raise ValueError("The probabilities in the distribution should add up to 1!")
        return rng.choices(list(value_or_distribution), weights=list(value_or_distribution.values()), k=1)[0]

    machines = tuple(Machine(f"m{i}") for i in range(n_machines))
    jobs = []
    for i in range(n_jobs):
        visited = rng.sample(machines, k=round(draw(relative_op_amount) * n_machines))
        rng.shuffle(visited)
        jobs.append(Job(f"job{i}", tuple(Operation(f"op{j}", f"job{i}", machine, draw(op_duration)) for j, machine in enumerate(visited))))
    return JobShopSchedulingProblemInstance(instance_name, machines, tuple(jobs))


# ---- JSON wire format of the datatypes above (reference: queasars/job_shop_scheduling/serialization.py:18-196) ------------
# Key names and nesting are the reference's, so files written by either side load in the other: a tuple is {"tuple": [...]},
# a dictionary {"dict": [item, ...]} with every item a two-element tuple, and each dataclass a dictionary of prefixed keys.

import json as _json

_WIRE_KEYS = {
    "machine": ("machine_name",),
    "operation": ("operation_name", "operation_job_name", "operation_machine", "operation_processing_duration"),
    "job": ("job_name", "job_operations"),
    "instance": ("jssp_instance_name", "jssp_instance_machines", "jssp_instance_jobs"),
    "unscheduled": ("unscheduled_operation",),
    "scheduled": ("scheduled_operation", "scheduled_start_time"),
    "result": ("jssp_result_problem_instance", "jssp_result_schedule"),
}


def jssp_to_wire(obj):
    """The JSON-ready form of a JSSP object (or of tuples / lists / dictionaries of them)."""
    if isinstance(obj, tuple):
        return {"tuple": [jssp_to_wire(x) for x in obj]}
    if isinstance(obj, list):
        return [jssp_to_wire(x) for x in obj]
    if isinstance(obj, dict):
        return {"dict": [jssp_to_wire(item) for item in obj.items()]}
    fields = None
    if isinstance(obj, Machine):
        kind, fields = "machine", (obj.name,)
    elif isinstance(obj, Operation):
        kind, fields = "operation", (obj.name, obj.job_name, obj.machine, obj.processing_duration)
    elif isinstance(obj, Job):
        kind, fields = "job", (obj.name, obj.operations)
    elif isinstance(obj, JobShopSchedulingProblemInstance):
        kind, fields = "instance", (obj.name, obj.machines, obj.jobs)
    elif isinstance(obj, UnscheduledOperation):
        kind, fields = "unscheduled", (obj.operation,)
    elif isinstance(obj, ScheduledOperation):
        kind, fields = "scheduled", (obj.operation, obj.start_time)
    elif isinstance(obj, JobShopSchedulingResult):
        kind, fields = "result", (obj.problem_instance, obj.schedule)
    if fields is None:
        return obj
    return {key: jssp_to_wire(value) for key, value in zip(_WIRE_KEYS[kind], fields)}


def _jssp_from_wire_dict(d: dict):
    """``object_hook`` of the decoder: called innermost first, so the values of ``d`` are decoded already."""
    if len(d) == 1 and "tuple" in d:
        return tuple(d["tuple"])
    if len(d) == 1 and "dict" in d:
        return {key: value for key, value in d["dict"]}
    makers = {
        "machine": Machine, "operation": Operation, "job": Job, "instance": JobShopSchedulingProblemInstance,
        "unscheduled": UnscheduledOperation, "scheduled": ScheduledOperation, "result": JobShopSchedulingResult,
    }
    for kind, keys in _WIRE_KEYS.items():
        if any(key in d for key in keys):
            missing = [key for key in keys if key not in d]
            if missing:
                raise ValueError(f"JSSP JSON: {kind} without {missing}")
            return makers[kind](*(d[key] for key in keys))
    return d


class JSSPJSONEncoder(_json.JSONEncoder):
    """``json.dumps(obj, cls=JSSPJSONEncoder)`` for Machine, Operation, Job, JobShopSchedulingProblemInstance,
    (Un)ScheduledOperation and JobShopSchedulingResult."""

    def iterencode(self, o, _one_shot=False):
        # (tuples and dictionaries keyed by objects never reach default(): the whole object is translated first;
        # json.dumps and json.dump both come through here, once)
        return super().iterencode(jssp_to_wire(o), _one_shot)


class JSSPJSONDecoder(_json.JSONDecoder):
    """``json.loads(text, cls=JSSPJSONDecoder)``."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("object_hook", None)
        super().__init__(*args, object_hook=_jssp_from_wire_dict, **kwargs)
