"""Shared test helpers: workload construction and access to the oracles (tests only)."""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from oracle import statevector_oracle as so
from queasars_amd.ir import CircuitIR, PauliOperator
from queasars_amd.workloads import population_circuits, random_ising_operator, random_pauli_operator  # noqa: F401

ROOT = Path(__file__).resolve().parent.parent


def oracle_state(circuit: CircuitIR, params) -> np.ndarray:
    return so.simulate(circuit.n_qubits, circuit.bound_ops(params))


def oracle_expectation(circuit: CircuitIR, params, operator: PauliOperator) -> float:
    state = oracle_state(circuit, params)
    return so.pauli_expectation(state, operator.x_mask.tolist(), operator.z_mask.tolist(), operator.coeffs.tolist()).real


def inverse_circuit(circuit: CircuitIR, params) -> CircuitIR:
    """Bound circuit that undoes ``circuit``: gates reversed, u(t,p,l)^-1 = u(-t,-l,-p)."""
    inv = CircuitIR(circuit.n_qubits)
    for kind, target, control, theta, phi, lam in reversed(circuit.bound_ops(params)):
        if kind == 0:
            inv.id(target)
        elif kind == 1:
            inv.u(-theta, -lam, -phi, target)
        else:
            inv.cu3(-theta, -lam, -phi, control, target)
    return inv


def bound_copy(circuit: CircuitIR, params) -> CircuitIR:
    out = CircuitIR(circuit.n_qubits)
    for kind, target, control, theta, phi, lam in circuit.bound_ops(params):
        if kind == 0:
            out.id(target)
        elif kind == 1:
            out.u(theta, phi, lam, target)
        else:
            out.cu3(theta, phi, lam, control, target)
    return out


class COracle:
    def __init__(self, lib):
        self.lib = lib
        lib.qsvo_simulate.restype = C.c_int
        lib.qsvo_evaluate.restype = C.c_double
        lib.qsvo_diagonal_expectation_table.restype = C.c_double
        lib.qsvo_max_threads.restype = C.c_int

    @staticmethod
    def _arrays(circuit: CircuitIR, params):
        ops = circuit.bound_ops(params)
        kinds = np.asarray([o[0] for o in ops], dtype=np.int32)
        targets = np.asarray([o[1] for o in ops], dtype=np.int32)
        controls = np.asarray([o[2] for o in ops], dtype=np.int32)
        angles = np.asarray([[o[3], o[4], o[5]] for o in ops], dtype=np.float64).reshape(-1)
        return kinds, targets, controls, angles

    def simulate(self, circuit: CircuitIR, params) -> np.ndarray:
        kinds, targets, controls, angles = self._arrays(circuit, params)
        state = np.zeros(2 << circuit.n_qubits, dtype=np.float64)
        rc = self.lib.qsvo_simulate(
            circuit.n_qubits, len(kinds), kinds.ctypes, targets.ctypes, controls.ctypes, angles.ctypes, state.ctypes, 1
        )
        assert rc == 0, rc
        return state.view(np.complex128)

    def diagonal_table(self, operator: PauliOperator) -> np.ndarray:
        table = np.zeros(1 << operator.num_qubits, dtype=np.float64)
        z = np.ascontiguousarray(operator.z_mask)
        c = np.ascontiguousarray(operator.coeffs.real)
        self.lib.qsvo_diagonal_table(operator.num_qubits, len(operator), z.ctypes, c.ctypes, table.ctypes)
        return table

    def evaluate(self, circuit: CircuitIR, params, operator: PauliOperator, table=None, scratch=None) -> float:
        kinds, targets, controls, angles = self._arrays(circuit, params)
        if scratch is None:
            scratch = np.zeros(2 << circuit.n_qubits, dtype=np.float64)
        x = np.ascontiguousarray(operator.x_mask)
        z = np.ascontiguousarray(operator.z_mask)
        cre = np.ascontiguousarray(operator.coeffs.real)
        cim = np.ascontiguousarray(operator.coeffs.imag)
        return self.lib.qsvo_evaluate(
            circuit.n_qubits, len(kinds), kinds.ctypes, targets.ctypes, controls.ctypes, angles.ctypes, len(operator),
            x.ctypes, z.ctypes, cre.ctypes, cim.ctypes, table.ctypes if table is not None else None, scratch.ctypes,
        )


def host_cpu_share(limit: int = 16) -> int:
    """CPUs this process may really use: its affinity mask, the cgroup's quota, at most ``limit``.  (A GPU box shows all
    256 hardware threads of its host but grants a 16-CPU quota: OpenMP's default of one thread per visible CPU made an
    oracle evaluation 30 times slower there.)"""
    import os

    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            share = min(share, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(share, limit))


def load_c_oracle() -> COracle:
    so_path = ROOT / "oracle" / "libqsv_oracle.so"
    src = ROOT / "oracle" / "qsv_oracle.c"
    if not so_path.exists() or so_path.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    oracle = COracle(C.CDLL(str(so_path)))
    oracle.lib.qsvo_set_threads(host_cpu_share())
    return oracle


def textbook_cases():
    """Circuits whose expectation values are textbook facts, independent of any simulator: (name, circuit, {label: value}).
    Pauli labels in Qiskit's order (rightmost character = qubit 0).  They pin the roles of a cu3's control and target, the sign
    conventions of U's phases and the Y phase of the expectation -- the semantics DESIGN.md section 2 lists as taken from
    upstream documentation."""
    pi = np.pi
    h = (pi / 2, 0.0, pi)  # U(pi/2, 0, pi) = H
    x = (pi, 0.0, pi)      # U(pi, 0, pi) = X
    cases = []
    bell = CircuitIR(2).u(*h, 0).cu3(*x, 0, 1)  # (|00> + |11>) / sqrt 2
    cases.append(("Bell", bell, {"ZZ": 1.0, "XX": 1.0, "YY": -1.0, "ZI": 0.0, "IZ": 0.0, "XY": 0.0, "XI": 0.0}))
    ghz = CircuitIR(3).u(*h, 0).cu3(*x, 0, 1).cu3(*x, 1, 2)  # (|000> + |111>) / sqrt 2: Mermin's signs
    cases.append(("GHZ", ghz, {"XXX": 1.0, "XYY": -1.0, "YXY": -1.0, "YYX": -1.0, "ZZI": 1.0, "IZZ": 1.0, "ZIZ": 1.0, "ZII": 0.0}))
    lam = 0.83
    cphase = CircuitIR(2).u(*h, 0).u(*h, 1).cu3(0.0, 0.0, lam, 0, 1)  # (|00> + |01> + |10> + e^{i lam} |11>) / 2
    cases.append(("controlled phase", cphase, {"XI": (1 + np.cos(lam)) / 2, "IX": (1 + np.cos(lam)) / 2, "YI": np.sin(lam) / 2,
                                               "ZI": 0.0, "ZZ": 0.0, "XX": (1 + np.cos(lam)) / 2}))
    theta, phi = 1.1, 0.45
    bloch = CircuitIR(1).u(theta, phi, 0.3, 0)  # cos(theta/2)|0> + e^{i phi} sin(theta/2)|1>: the Bloch vector
    cases.append(("Bloch vector", bloch, {"X": np.sin(theta) * np.cos(phi), "Y": np.sin(theta) * np.sin(phi), "Z": np.cos(theta)}))
    # a controlled rotation whose control is |1>: the target carries the rotated state, the control is untouched
    crot = CircuitIR(2).u(*x, 1).cu3(theta, phi, -0.2, 1, 0)
    cases.append(("controlled rotation", crot, {"IX": np.sin(theta) * np.cos(phi), "IY": np.sin(theta) * np.sin(phi),
                                                "IZ": np.cos(theta), "ZI": -1.0, "ZZ": -np.cos(theta)}))
    return cases
