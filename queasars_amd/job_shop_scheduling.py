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


@dataclass(frozen=True)
class JobShopSchedulingProblemInstance:
    name: str
    machines: tuple[Machine, ...]
    jobs: tuple[Job, ...]

    def __post_init__(self):
        if len({m.name for m in self.machines}) != len(self.machines):
            raise JobShopSchedulingProblemException("Machine names must be unique!")
        if len({j.name for j in self.jobs}) != len(self.jobs):
            raise JobShopSchedulingProblemException("Job names must be unique!")
        for job in self.jobs:
            if any(op.machine not in self.machines for op in job.operations):
                raise JobShopSchedulingProblemException("A job uses a machine the instance does not have!")


@dataclass(frozen=True)
class JobShopSchedulingResult:
    """Start time per operation (``None`` = the variable's qubits do not hold a valid domain-wall state)."""

    problem_instance: JobShopSchedulingProblemInstance
    start_times: dict[Operation, Optional[int]]

    @property
    def is_valid(self) -> bool:
        if any(t is None for t in self.start_times.values()):
            return False
        per_machine: dict[Machine, list[tuple[int, int]]] = {m: [] for m in self.problem_instance.machines}
        for job in self.problem_instance.jobs:
            previous_end = None
            for op in job.operations:
                start = self.start_times[op]
                if previous_end is not None and start < previous_end:
                    return False
                previous_end = start + op.processing_duration
                per_machine[op.machine].append((start, previous_end))
        for spans in per_machine.values():
            spans.sort()
            if any(b[0] < a[1] for a, b in zip(spans, spans[1:])):
                return False
        return True

    @property
    def makespan(self) -> Optional[int]:
        if not self.is_valid:
            return None
        return max(self.start_times[job.operations[-1]] + job.operations[-1].processing_duration for job in self.problem_instance.jobs)


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
        return JobShopSchedulingResult(
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
