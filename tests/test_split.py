"""Register splitting (csrc/split.hpp) on the CPU: the two virtual circuits the scheduler builds, simulated with the
oracle's own gate application and contracted in NumPy, reproduce the oracle's state of the whole circuit."""
import ctypes as C

import numpy as np
import pytest

import helpers
from oracle import statevector_oracle as so
from queasars_amd import _lib
from queasars_amd.evqe import EVQEPopulation
from queasars_amd.ir import QSV_OP_DTYPE, CircuitIR

FIXED = {
    -2: np.array([[1, 0], [0, 0]], dtype=np.complex128),
    -3: np.array([[1, 0], [1, 0]], dtype=np.complex128),
    -4: np.array([[0, 1], [1, 0]], dtype=np.complex128),
}


def describe(circuit: CircuitIR, max_side: int):
    lib = _lib.load()
    ops = circuit.packed()
    cap = 4 * len(ops) + 64
    out_a, out_b = np.zeros(cap, dtype=QSV_OP_DTYPE), np.zeros(cap, dtype=QSV_OP_DTYPE)
    na, nb, mask = C.c_int(0), C.c_int(0), C.c_uint64(0)
    k = lib.qsv_split_describe(circuit.n_qubits, len(ops), _lib.as_ptr(ops), max_side, C.byref(mask), _lib.as_ptr(out_a), cap,
                               C.byref(na), _lib.as_ptr(out_b), cap, C.byref(nb))
    assert k >= -1, k
    if k < 0:
        return None
    return k, int(mask.value), out_a[: na.value], out_b[: nb.value]


def run_virtual(n_virtual: int, ops, params) -> np.ndarray:
    state = so.zero_state(n_virtual)
    for o in ops:
        if o["p_theta"] < -1:
            m = FIXED[int(o["p_theta"])]
        else:
            angles = [params[int(o[p])] if o[p] >= 0 else float(o[v]) for p, v in (("p_theta", "theta"), ("p_phi", "phi"), ("p_lambda", "lam"))]
            m = so.u_matrix(*angles)
        if o["kind"] == so.U:
            state = so._apply_1q(state, n_virtual, int(o["target"]), m)
        else:
            state = so._apply_controlled_1q(state, n_virtual, int(o["control"]), int(o["target"]), m)
    return state


def contract(n: int, k: int, mask_a: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    idx = np.arange(1 << n, dtype=np.int64)
    ia, ib, ca, cb = np.zeros_like(idx), np.zeros_like(idx), 0, 0
    for q in range(n):
        if (mask_a >> q) & 1:
            ia |= ((idx >> q) & 1) << ca
            ca += 1
        else:
            ib |= ((idx >> q) & 1) << cb
            cb += 1
    psi = np.zeros(1 << n, dtype=np.complex128)
    for kappa in range(1 << k):
        psi += a[(kappa << ca) + ia] * b[(kappa << cb) + ib]
    return psi


def check(circuit: CircuitIR, params, max_side: int):
    got = describe(circuit, max_side)
    if got is None:
        return None
    k, mask_a, ops_a, ops_b = got
    n = circuit.n_qubits
    na = bin(mask_a).count("1")
    assert na + k <= max_side and n - na + k <= max_side
    a = run_virtual(na + k, ops_a, params)
    b = run_virtual(n - na + k, ops_b, params)
    psi = contract(n, k, mask_a, a, b)
    ref = helpers.oracle_state(circuit, params)
    assert np.abs(psi - ref).max() < 1e-13
    return k


@pytest.mark.parametrize("n,layers,max_side", [(10, 3, 7), (12, 4, 8), (14, 4, 9), (14, 6, 9), (16, 4, 10), (12, 6, 11), (14, 7, 12),
                                               (16, 6, 13)])
def test_virtual_circuits_reproduce_the_state(n, layers, max_side):
    population = EVQEPopulation.random_population(n, layers, 24, True, 5 + n)
    seen = []
    for ind in population.individuals:
        k = check(ind.get_parameterized_quantum_circuit(), list(ind.parameter_values), max_side)
        if k is not None:
            seen.append(k)
    assert seen, "no circuit of the population was split"
    if layers >= 4:
        assert max(seen) >= 1, "only fully separable circuits: the key construction was not exercised"
    if 2 * max_side >= n + 8 and layers >= 6:  # (room for four keys on both sides)
        assert max(seen) >= 4, f"no circuit with four or five keys (16 / 32 product terms): {seen}"


def test_keys_and_projections_by_hand():
    """One control used twice in a row (one key), then rotated and used again (a second key); a control that is later
    targeted across the cut; a gate whose control nobody touched (dropped)."""
    c = CircuitIR(6)
    for q in range(6):
        c.u(0.3 + 0.1 * q, 0.2 * q, -0.1 * q, q)
    c.cu3(0.7, 0.1, 0.2, 0, 3).cu3(0.5, -0.3, 0.4, 0, 4)      # key (0, e): two cross gates, one key
    c.u(1.1, 0.2, 0.3, 0).cu3(0.9, 0.8, -0.7, 0, 5)             # qubit 0 rotated: a new key
    c.cu3(0.4, 0.5, 0.6, 3, 1).u(0.2, 0.1, 0.0, 3).cu3(0.3, 0.2, 0.1, 4, 3)
    c.cu3(0.6, 0.1, 0.1, 1, 2).cu3(0.2, 0.3, 0.4, 4, 5)
    for max_side in (4, 5):
        got = describe(c, max_side)
        if got is not None:
            assert check(c, [], max_side) == got[0]
    fresh = CircuitIR(6).cu3(0.5, 0.1, 0.2, 0, 1)  # control still |0>: identity, the register is a product
    got = describe(fresh, 4)
    assert got is not None and got[0] == 0 and len(got[2]) + len(got[3]) == 0


def test_no_split_when_the_register_is_well_entangled_or_fits_a_tile():
    n = 10
    ring = CircuitIR(n)
    for q in range(n):
        ring.u(0.3, 0.1, 0.2, q)
    for rep in range(3):
        for q in range(n):
            ring.cu3(0.5 + rep, 0.1, 0.2, q, (q + 1 + rep) % n)
            ring.u(0.1 * q, 0.2, 0.3, q)
    assert describe(ring, 6) is None
    assert describe(ring, 10) is None  # n <= max_side: the whole register is one tile anyway


def _keys_of(circuit: CircuitIR):
    """The splitter's keys restated: a key is one control qubit between two gates that target it; a cu3 whose control
    nobody has targeted yet is the identity.  Returns [(control, {targets})]."""
    touched, epoch, keys = set(), {}, {}
    for op in circuit.packed():
        kind, target, control = int(op["kind"]), int(op["target"]), int(op["control"])
        if kind == 0:
            continue
        if kind == 2:
            if control not in touched:
                continue
            keys.setdefault((control, epoch.get(control, 0)), set()).add(target)
        touched.add(target)
        epoch[target] = epoch.get(target, 0) + 1
    return [(c, ts) for (c, _), ts in keys.items()]


@pytest.mark.parametrize("n,layers,max_side", [(8, 3, 5), (9, 4, 6), (10, 4, 6), (10, 5, 7), (11, 3, 6), (10, 5, 9), (11, 6, 10),
                                               (12, 5, 10), (12, 7, 11), (13, 6, 11)])
def test_the_partition_cuts_as_few_keys_as_any(n, layers, max_side):
    """Against brute force over all 2^(n-1) bipartitions: the splitter's number of keys is the minimum over the partitions
    whose virtual circuits fit the size limit, and it reports no split form exactly when there is none with at most five
    keys (up to three keys: the pruned enumeration -- articulation points, one enumeration for several limits --; four and
    five: branch and bound over the qubits; both exact)."""
    population = EVQEPopulation.random_population(n, layers, 40, True, 300 + n + layers)
    outcomes = set()
    for individual in population.individuals:
        circuit = individual.get_parameterized_quantum_circuit()
        keys = _keys_of(circuit)
        best = None
        for a_mask in range(1, 1 << (n - 1)):  # qubit n - 1 always on side B: every bipartition once
            cut = sum(1 for c, ts in keys if any(((a_mask >> c) & 1) != ((a_mask >> t) & 1) for t in ts))
            size_a = bin(a_mask).count("1")
            if cut <= 5 and size_a + cut <= max_side and n - size_a + cut <= max_side and (best is None or cut < best):
                best = cut
        got = describe(circuit, max_side)
        assert (got[0] if got is not None else None) == best, (keys, got and got[:2], best)
        outcomes.add(best)
    assert len(outcomes) >= 2, outcomes  # (the populations exercise several outcomes)
