"""GPU tests of the BASELINE.json configurations at (or near) their full sizes, of the committed golden fixtures
through the two batch entry points of the C ABI, and of the remaining ABI surface.  Run on the MI355X box with -m gpu.

Tolerances (stated where used): fp64 expectation values within 1e-10 of the oracle (north_star); fp32 expectation
values within FP32_REL * sum_k |c_k| of the fp64 value (fp32 carries 2^-24 = 6e-8 per rounding; a four-layer circuit
applies a few dozen butterflies to every amplitude and the reductions accumulate in fp64).
"""

import ctypes as C
import os
import json
from pathlib import Path

import numpy as np
import pytest

import helpers
from queasars_amd import _lib
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator, OperatorSamplerCircuitEvaluator, StatevectorDevice
from queasars_amd.ir import QSV_OP_DTYPE, CircuitIR, PauliOperator

pytestmark = pytest.mark.gpu

EXP_TOL = 1e-10
FP32_REL = 2e-6
GOLDEN = Path(__file__).resolve().parent / "golden" / "evqe_small.json"


# ---- raw C ABI helpers (no StatevectorDevice: these tests call libqsv the way INTEGRATION.md's stub does) -----------


class RawHandle:
    def __init__(self, n_qubits, dtype=_lib.QSV_F64):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        rc = self.lib.qsv_create(n_qubits, dtype, 0, None, C.byref(self.h))
        assert rc == _lib.QSV_OK, _lib.last_error(self.lib, None)

    def check(self, rc):
        assert rc == _lib.QSV_OK, _lib.last_error(self.lib, self.h)

    def set_operator(self, op: PauliOperator):
        x, z = np.ascontiguousarray(op.x_mask), np.ascontiguousarray(op.z_mask)
        cre, cim = np.ascontiguousarray(op.coeffs.real), np.ascontiguousarray(op.coeffs.imag)
        self.check(self.lib.qsv_set_operator(self.h, len(op), _lib.as_ptr(x), _lib.as_ptr(z), _lib.as_ptr(cre), _lib.as_ptr(cim)))

    def close(self):
        if self.h:
            self.lib.qsv_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def ops_array(rows, parameterised: bool):
    """qsv_op records of bound ``(kind, target, control, theta, phi, lam)`` rows: with literal angles, or with every
    angle of gate g taken from params[3g .. 3g+3) (and a literal that must then be ignored)."""
    arr = np.zeros(len(rows), dtype=QSV_OP_DTYPE)
    params = []
    for i, (kind, target, control, theta, phi, lam) in enumerate(rows):
        ctrl = 0xFF if control is None or control < 0 else control
        if parameterised and kind != 0:
            base = len(params)
            params += [theta, phi, lam]
            arr[i] = (kind, target, ctrl, 0, base, base + 1, base + 2, 99.0, 99.0, 99.0)
        else:
            arr[i] = (kind, target, ctrl, 0, -1, -1, -1, theta, phi, lam)
    return arr, np.asarray(params, dtype=np.float64)


def golden_cases():
    return json.loads(GOLDEN.read_text())["cases"]


# ---- (iii) committed golden fixtures through qsv_eval_batch and qsv_eval_circuits --------------------------------


@pytest.mark.parametrize("parameterised", [False, True])
def test_golden_fixtures_through_qsv_eval_batch(parameterised):
    """Every case of tests/golden/evqe_small.json through ``qsv_eval_batch`` (op lists inline; replaces
    ``estimator.run(pubs)``, circuit_evaluation.py:204-215).  Cases of one qubit count share a handle and operator
    changes in between; the batch holds each case three times to exercise the inline-structure cache."""
    by_n = {}
    for case in golden_cases():
        by_n.setdefault(case["n_qubits"], []).append(case)
    worst = 0.0
    for n, cases in sorted(by_n.items()):
        with RawHandle(n) as raw:
            for case in cases:
                raw.set_operator(PauliOperator(case["labels"], case["coeffs"]))
                ops, params = ops_array([tuple(o) for o in case["ops"]], parameterised)
                reps = 3
                all_ops = np.concatenate([ops] * reps)
                op_offsets = np.arange(reps + 1, dtype=np.int64) * len(ops)
                all_params = np.concatenate([params] * reps) if params.size else np.zeros(1)
                param_offsets = np.arange(reps + 1, dtype=np.int64) * len(params)
                out = np.full(reps, np.nan)
                raw.check(raw.lib.qsv_eval_batch(raw.h, reps, _lib.as_ptr(op_offsets), _lib.as_ptr(all_ops),
                                                 _lib.as_ptr(param_offsets), _lib.as_ptr(all_params), _lib.as_ptr(out)))
                assert out[0] == out[1] == out[2]
                worst = max(worst, abs(out[0] - case["expectation"]))
    assert worst < EXP_TOL


def test_golden_fixtures_through_qsv_eval_circuits_and_statevector():
    """The same fixtures through ``qsv_circuit_create`` + ``qsv_eval_circuits`` (the call INTEGRATION.md's ctypes stub
    makes), all cases of one qubit count in ONE call where they share an operator... they do not, so one call per
    case plus one mixed call with gaps between the parameter vectors; stored amplitudes through ``qsv_statevector``."""
    worst_e = worst_a = 0.0
    for case in golden_cases():
        n = case["n_qubits"]
        with RawHandle(n) as raw:
            raw.set_operator(PauliOperator(case["labels"], case["coeffs"]))
            rows = [tuple(o) for o in case["ops"]]
            lit, _ = ops_array(rows, False)
            par, params = ops_array(rows, True)
            cid_lit, cid_par = C.c_int(0), C.c_int(0)
            raw.check(raw.lib.qsv_circuit_create(raw.h, len(lit), _lib.as_ptr(lit), 0, C.byref(cid_lit)))
            raw.check(raw.lib.qsv_circuit_create(raw.h, len(par), _lib.as_ptr(par), len(params), C.byref(cid_par)))
            # three evaluations; the parameter vectors sit in a buffer with unused gaps between them
            ids = np.asarray([cid_par.value, cid_lit.value, cid_par.value], dtype=np.int32)
            buf = np.full(2 * len(params) + 7, 1234.5)
            buf[: len(params)] = params
            buf[len(params) + 5 : 2 * len(params) + 5] = params
            offsets = np.asarray([0, len(params), len(params) + 5, 2 * len(params) + 5], dtype=np.int64)
            # evaluation 1 (literal circuit) declares the gap as its vector: extra values are ignored
            out = np.full(3, np.nan)
            raw.check(raw.lib.qsv_eval_circuits(raw.h, 3, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(buf), _lib.as_ptr(out)))
            worst_e = max(worst_e, float(np.abs(out - case["expectation"]).max()))
            if "state_re" in case:
                state = np.empty(2 << n)
                raw.check(raw.lib.qsv_statevector(raw.h, cid_par.value, _lib.as_ptr(params) if params.size else None,
                                                  len(params), _lib.as_ptr(state)))
                want = np.asarray(case["state_re"]) + 1j * np.asarray(case["state_im"])
                worst_a = max(worst_a, float(np.abs(state.view(np.complex128) - want).max()))
            raw.check(raw.lib.qsv_circuit_destroy(raw.h, cid_lit.value))
            assert raw.lib.qsv_circuit_destroy(raw.h, cid_lit.value) == _lib.QSV_E_ARG  # already gone
    assert worst_e < EXP_TOL and worst_a < 1e-12


def test_eval_batch_survives_more_structures_than_its_cache_holds():
    """``qsv_eval_batch`` keeps at most 4096 inline structures registered.  Push more distinct structures than that,
    within one call and across calls: every result must still be right (the round-1 library freed plans that earlier
    evaluations of the same call pointed at)."""
    n = 5
    op = helpers.random_pauli_operator(n, 6, seed=3)
    rng = np.random.default_rng(0)

    def batch(count):
        angles = rng.uniform(0, 2 * np.pi, size=(count, 6))
        ops = np.zeros(3 * count, dtype=QSV_OP_DTYPE)
        for i, a in enumerate(angles):
            ops[3 * i] = (1, 0, 0xFF, 0, -1, -1, -1, a[0], a[1], a[2])
            ops[3 * i + 1] = (1, 3, 0xFF, 0, -1, -1, -1, a[3], 0.1, 0.2)
            ops[3 * i + 2] = (2, 2, 0, 0, -1, -1, -1, a[4], a[5], 0.3)  # cu3 control 0 -> target 2
        return angles, ops

    def reference(a):
        c = CircuitIR(n).u(a[0], a[1], a[2], 0).u(a[3], 0.1, 0.2, 3).cu3(a[4], a[5], 0.3, 0, 2)
        return helpers.oracle_expectation(c, [], op)

    with RawHandle(n) as raw:
        raw.set_operator(op)
        for count in (3000, 3000, 5000, 10):  # the cache overflows between calls 2 and 3, and inside call 3
            angles, ops = batch(count)
            op_offsets = np.arange(count + 1, dtype=np.int64) * 3
            param_offsets = np.zeros(count + 1, dtype=np.int64)
            out = np.full(count, np.nan)
            raw.check(raw.lib.qsv_eval_batch(raw.h, count, _lib.as_ptr(op_offsets), _lib.as_ptr(ops), _lib.as_ptr(param_offsets),
                                             None, _lib.as_ptr(out)))
            for i in list(range(0, count, max(1, count // 40))) + [count - 1]:
                assert abs(out[i] - reference(angles[i])) < EXP_TOL


def test_sample_single_and_foreign_stream():
    """``qsv_sample`` (one circuit) and ``qsv_set_stream`` (launch on a caller-owned HIP stream, here one of torch's)."""
    import torch

    n = 9
    _, circuits, params = helpers.population_circuits(n, 2, 2, seed=5)
    op = helpers.random_ising_operator(n, seed=7)
    with RawHandle(n) as raw:
        raw.set_operator(op)
        ops = circuits[0].packed()
        cid = C.c_int(0)
        raw.check(raw.lib.qsv_circuit_create(raw.h, len(ops), _lib.as_ptr(ops), circuits[0].num_parameters, C.byref(cid)))
        p = np.asarray(params[0], dtype=np.float64)
        offsets = np.asarray([0, len(p)], dtype=np.int64)
        ids = np.asarray([cid.value], dtype=np.int32)
        before = np.zeros(1)
        raw.check(raw.lib.qsv_eval_circuits(raw.h, 1, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(p), _lib.as_ptr(before)))
        stream = torch.cuda.Stream()
        raw.check(raw.lib.qsv_set_stream(raw.h, C.c_void_p(stream.cuda_stream)))
        after = np.zeros(1)
        raw.check(raw.lib.qsv_eval_circuits(raw.h, 1, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(p), _lib.as_ptr(after)))
        assert after[0] == before[0] and abs(after[0] - helpers.oracle_expectation(circuits[0], params[0], op)) < EXP_TOL
        shots = 50_000
        states = np.zeros(shots, dtype=np.uint64)
        raw.check(raw.lib.qsv_sample(raw.h, cid.value, _lib.as_ptr(p), len(p), shots, C.c_uint64(77), _lib.as_ptr(states)))
        again = np.zeros(shots, dtype=np.uint64)
        raw.check(raw.lib.qsv_sample(raw.h, cid.value, _lib.as_ptr(p), len(p), shots, C.c_uint64(77), _lib.as_ptr(again)))
        assert np.array_equal(states, again)
        probs = np.abs(helpers.oracle_state(circuits[0], params[0])) ** 2
        freq = np.bincount(states.astype(np.int64), minlength=1 << n) / shots
        assert np.all(np.abs(freq - probs) < 6 * np.sqrt(probs * (1 - probs) / shots) + 2.0 / shots)
        raw.check(raw.lib.qsv_set_stream(raw.h, None))  # back to a stream of the library's own
        raw.check(raw.lib.qsv_eval_circuits(raw.h, 1, _lib.as_ptr(ids), _lib.as_ptr(offsets), _lib.as_ptr(p), _lib.as_ptr(after)))
        assert after[0] == before[0]
        # a second begin on the thread that holds an open batch is refused instead of waiting for itself
        counts = np.asarray([len(p)], dtype=np.int64)
        raw.check(raw.lib.qsv_eval_begin(raw.h, 1, _lib.as_ptr(ids), _lib.as_ptr(counts)))
        assert raw.lib.qsv_eval_begin(raw.h, 1, _lib.as_ptr(ids), _lib.as_ptr(counts)) == _lib.QSV_E_STATE
        raw.check(raw.lib.qsv_eval_push(raw.h, 0, 1, _lib.as_ptr(p)))
        raw.check(raw.lib.qsv_eval_end(raw.h, _lib.as_ptr(after)))
        assert after[0] == before[0]


# ---- (ii) BASELINE config 3's workload on one GPU, against the C oracle -------------------------------------------


def test_config3_n24_population_against_the_c_oracle(c_oracle):
    """n = 24, L = 4, the 300-term Ising operator of default_rng(2024): individuals 0, 1, 100 and 255 of the
    256-individual population (seed 0) directly against the plain-C oracle, |dE| <= 1e-10."""
    n = 24
    _, circuits, params = helpers.population_circuits(n, 4, 256, seed=0)
    pick = [0, 1, 100, 255]
    cs, ps = [circuits[i] for i in pick], [params[i] for i in pick]
    op = helpers.random_ising_operator(n, seed=2024)
    assert len(op) == 300
    got = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(cs, ps))
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    ref = np.asarray([c_oracle.evaluate(c, p, op, table, scratch) for c, p in zip(cs, ps)])
    assert np.abs(got - ref).max() < EXP_TOL


# ---- (i) BASELINE config 5: general 500-term operator, fp64 and fp32 ----------------------------------------------


@pytest.mark.parametrize("n_qubits", [20, 23])
def test_config5_general_operator_against_the_c_oracle(n_qubits, c_oracle):
    """The general-operator path (x-mask groups, pair kernels) with config 5's operator family -- 500 random Pauli
    strings over {I,X,Y,Z}, coefficients uniform(-1,1), default_rng(2028) -- at sizes the C oracle finishes in
    seconds: fp64 within 1e-10, fp32 within FP32_REL * sum |c_k| of the oracle."""
    _, circuits, params = helpers.population_circuits(n_qubits, 4, 2, seed=0)
    op = helpers.random_pauli_operator(n_qubits, 500, seed=2028)
    ref = np.asarray([c_oracle.evaluate(c, p, op) for c, p in zip(circuits, params)])
    got64 = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
    assert np.abs(got64 - ref).max() < EXP_TOL
    got32 = np.asarray(OperatorCircuitEvaluator(op, dtype="fp32").evaluate_circuits(circuits, params))
    assert np.abs(got32 - ref).max() < FP32_REL * float(np.abs(op.coeffs).sum())


def test_config5_n28_fp32_against_fp64():
    """BASELINE config 5 at full size: n = 28, one genome (L = 4, seed 0), 500 Pauli strings (default_rng(2028)), fp64
    (4 GiB state) and fp32 (2 GiB).  No CPU oracle at this size: fp32 against fp64 within
    FP32_REL * sum |c_k|, and on the fp64 path size-independent properties -- the two halves of the operator add up to
    the whole (1e-10), <I> = 1, and a basis-state circuit gives the diagonal terms' exact value."""
    n = 28
    _, circuits, params = helpers.population_circuits(n, 4, 1, seed=0)
    op = helpers.random_pauli_operator(n, 500, seed=2028)
    dev = StatevectorDevice(n)
    whole = OperatorCircuitEvaluator(op, statevector_device=dev).evaluate_circuits(circuits, params)[0]
    labels, coeffs = op.labels, op.coeffs
    first = OperatorCircuitEvaluator(PauliOperator(labels[:250], coeffs[:250]), statevector_device=dev).evaluate_circuits(circuits, params)[0]
    second = OperatorCircuitEvaluator(PauliOperator(labels[250:], coeffs[250:]), statevector_device=dev).evaluate_circuits(circuits, params)[0]
    assert abs(whole - (first + second)) < EXP_TOL
    ident = PauliOperator(["I" * n], [1.0])
    assert abs(OperatorCircuitEvaluator(ident, statevector_device=dev).evaluate_circuits(circuits, params)[0] - 1.0) < 1e-11
    # basis state |b>: <P> = 0 for every string with an X or Y, +-1 for I/Z strings
    bits = 0b1011_0000_1111_0101_0011_1100_1010
    flip = CircuitIR(n)
    for q in range(n):
        if (bits >> q) & 1:
            flip.u(np.pi, 0.0, np.pi, q)
    mixed = PauliOperator(["Z" * n, "I" * (n - 3) + "ZIZ", "X" + "I" * (n - 1), "I" * (n - 1) + "Y"], [0.5, -2.0, 7.0, 11.0])
    want = 0.5 * (-1) ** bin(bits).count("1") - 2.0 * (-1) ** bin(bits & 0b101).count("1")
    assert abs(OperatorCircuitEvaluator(mixed, statevector_device=dev).evaluate_circuits([flip], [[]])[0] - want) < 1e-12
    dev.close()
    single = OperatorCircuitEvaluator(op, dtype="fp32").evaluate_circuits(circuits, params)[0]
    assert abs(single - whole) < FP32_REL * float(np.abs(coeffs).sum())


# ---- (v) fp32 expectation values against fp64 ---------------------------------------------------------------------


@pytest.mark.parametrize("n_qubits", [12, 16, 20])
def test_fp32_expectation_error_is_bounded(n_qubits):
    """fp32 states, fp64 accumulation: |<H>_fp32 - <H>_fp64| <= FP32_REL * sum |c_k| for the diagonal (fused) path and
    for the general path, on four individuals each."""
    _, circuits, params = helpers.population_circuits(n_qubits, 4, 4, seed=n_qubits)
    for op in (helpers.random_ising_operator(n_qubits, seed=5), helpers.random_pauli_operator(n_qubits, 40, seed=6)):
        e64 = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
        e32 = np.asarray(OperatorCircuitEvaluator(op, dtype="fp32").evaluate_circuits(circuits, params))
        assert np.abs(e32 - e64).max() < FP32_REL * float(np.abs(op.coeffs).sum())
        assert np.abs(e32 - e64).max() > 0  # fp32 really ran


# ---- (iv) the 3 x 3 JSSP instance (18 qubits) ----------------------------------------------------------------------


def test_config4_three_by_three_jssp_end_to_end():
    """BASELINE.json words config 4 as a 3-jobs x 3-machines instance: unit durations, makespan limit 5, 18 qubits.
    EVQE (sampler + CVaR 0.5, 512 shots) must end on a valid schedule of optimal makespan 3, and the energy it
    reports must be the exact energy of that schedule's basis state."""
    import jssp_instances as inst
    from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator
    from queasars_amd.evqe.solver import (
        SPSA, BestIndividualRelativeChangeTolerance, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration, SPSATerminationChecker,
    )
    from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

    enc = JSSPDomainWallHamiltonianEncoder(inst.three_by_three(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 18
    op = enc.get_problem_hamiltonian()
    evaluator = OperatorSamplerCircuitEvaluator(512, op, alpha=0.5, seed=0)
    cfg = EVQEMinimumEigensolverConfiguration(
        optimizer=SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True,
                       termination_checker=SPSATerminationChecker(0.01, 2)),
        population_size=10, max_generations=8, termination_criterion=BestIndividualRelativeChangeTolerance(0.01, 1),
        random_seed=0, n_initial_layers=2, randomize_initial_population_parameters=True,
        speciation_genetic_distance_threshold=1, use_tournament_selection=True, tournament_size=2,
        selection_alpha_penalty=0.15, selection_beta_penalty=0.02, parameter_search_probability=0.39,
        topological_search_probability=0.79, layer_removal_probability=0.02,
    )
    result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(evaluator)
    best = result.best_individual
    dev = evaluator.statevector_device
    probs = dev.probabilities(best.get_parameterized_quantum_circuit(), list(best.parameter_values))
    top = int(np.argmax(probs))
    schedule = enc.translate_result_bitstring(format(top, f"0{enc.n_qubits}b"))
    assert schedule.is_valid and schedule.makespan == 3
    # the energy of that basis state, exactly, through the estimator branch on the same device
    prep = CircuitIR(enc.n_qubits)
    for q in range(enc.n_qubits):
        if (top >> q) & 1:
            prep.u(np.pi, 0.0, np.pi, q)
    exact = OperatorCircuitEvaluator(op, statevector_device=dev).evaluate_circuits([prep], [[]])[0]
    assert abs(exact - 5.0625) < 1e-9
    assert result.eigenvalue >= exact - 1e-9 and result.eigenvalue < exact + 1.0


def test_three_by_three_contended_jssp_end_to_end():
    """A 3 x 3 instance whose all-zero state is NOT a schedule (two jobs start on the same machine; optimum: makespan
    5, energy 36.708 by exhaustive search over the 2^18 basis states).  EVQE with the notebook's settings must end on
    a valid schedule -- no penalty term active, so an energy far below the penalties (275 / 319)."""
    import jssp_instances as inst
    from queasars_amd.circuit_evaluation import OperatorSamplerCircuitEvaluator
    from queasars_amd.evqe.solver import (
        SPSA, BestIndividualRelativeChangeTolerance, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration, SPSATerminationChecker,
    )
    from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder

    enc = JSSPDomainWallHamiltonianEncoder(inst.three_by_three_contended(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
    evaluator = OperatorSamplerCircuitEvaluator(512, enc.get_problem_hamiltonian(), alpha=0.5, seed=0)
    cfg = EVQEMinimumEigensolverConfiguration(
        optimizer=SPSA(maxiter=33, perturbation=0.35, learning_rate=0.43, trust_region=True,
                       termination_checker=SPSATerminationChecker(0.01, 2)),
        population_size=10, max_generations=8, termination_criterion=BestIndividualRelativeChangeTolerance(0.01, 1),
        random_seed=0, n_initial_layers=2, randomize_initial_population_parameters=True,
        speciation_genetic_distance_threshold=1, use_tournament_selection=True, tournament_size=2,
        selection_alpha_penalty=0.15, selection_beta_penalty=0.02, parameter_search_probability=0.39,
        topological_search_probability=0.79, layer_removal_probability=0.02,
    )
    result = EVQEMinimumEigensolver(cfg).compute_minimum_eigenvalue(evaluator)
    best = result.best_individual
    probs = evaluator.statevector_device.probabilities(best.get_parameterized_quantum_circuit(), list(best.parameter_values))
    schedule = enc.translate_result_bitstring(format(int(np.argmax(probs)), f"0{enc.n_qubits}b"))
    assert schedule.is_valid and schedule.makespan == 5
    assert 36.7 < result.eigenvalue < 100.0


def test_many_structures_registered_in_one_call():
    """``qsv_circuits_create``: a population of new structures is scheduled by several host threads in one call; a bad
    circuit among them fails the call and names the circuit."""
    n = 12
    _, circuits, params = helpers.population_circuits(n, 3, 40, seed=23)
    op = helpers.random_ising_operator(n, seed=3)
    with RawHandle(n) as raw:
        raw.set_operator(op)
        ops = np.concatenate([c.packed() for c in circuits])
        offsets = np.zeros(len(circuits) + 1, dtype=np.int64)
        np.cumsum([len(c) for c in circuits], out=offsets[1:])
        counts = np.asarray([c.num_parameters for c in circuits], dtype=np.int32)
        ids = np.zeros(len(circuits), dtype=np.int32)
        raw.check(raw.lib.qsv_circuits_create(raw.h, len(circuits), _lib.as_ptr(offsets), _lib.as_ptr(ops), _lib.as_ptr(counts), _lib.as_ptr(ids)))
        assert len(set(ids.tolist())) == len(circuits)
        flat = np.concatenate([np.asarray(p) for p in params])
        poffsets = np.zeros(len(circuits) + 1, dtype=np.int64)
        np.cumsum([len(p) for p in params], out=poffsets[1:])
        out = np.zeros(len(circuits))
        raw.check(raw.lib.qsv_eval_circuits(raw.h, len(circuits), _lib.as_ptr(ids), _lib.as_ptr(poffsets), _lib.as_ptr(flat), _lib.as_ptr(out)))
        ref = np.asarray([helpers.oracle_expectation(c, p, op) for c, p in zip(circuits[:6], params[:6])])
        assert np.abs(out[:6] - ref).max() < EXP_TOL
        bad = ops.copy()
        bad["target"][int(offsets[17])] = 99
        assert raw.lib.qsv_circuits_create(raw.h, len(circuits), _lib.as_ptr(offsets), _lib.as_ptr(bad), _lib.as_ptr(counts), _lib.as_ptr(ids)) == _lib.QSV_E_ARG
        assert b"circuit 17" in raw.lib.qsv_last_error(raw.h)
    # the Python layer takes the same path for a batch of unknown circuits: same bits as one-by-one registration
    ev = OperatorCircuitEvaluator(op)
    together = ev.evaluate_circuits(circuits, params)
    fresh = helpers.population_circuits(n, 3, 40, seed=23)[1]
    ev2 = OperatorCircuitEvaluator(op)
    one_by_one = [ev2.evaluate_circuits([c], [p])[0] for c, p in zip(fresh, params)]
    assert together == one_by_one and np.abs(np.asarray(together) - out).max() == 0.0


def test_native_coalescing_of_concurrent_one_circuit_calls():
    """The reference's calling pattern -- population_size threads, one circuit per call (selection.py:75-82) -- through
    ``qsv_eval_coalesced``: callers are merged inside the library; every caller gets exactly the value a batched call
    gives, and a caller's bad request fails that caller alone."""
    from concurrent.futures import ThreadPoolExecutor

    from queasars_amd.circuit_evaluation import CoalescingCircuitEvaluator

    n, P = 13, 48
    _, circuits, params = helpers.population_circuits(n, 3, P, seed=31)
    op = helpers.random_ising_operator(n, seed=9)
    ev = OperatorCircuitEvaluator(op)
    want = ev.evaluate_circuits(circuits, params)
    merged = CoalescingCircuitEvaluator(ev)
    assert merged._native
    with ThreadPoolExecutor(max_workers=P) as pool:
        for _ in range(5):
            got = list(pool.map(lambda j: merged.evaluate_circuits([circuits[j]], [params[j]])[0], range(P)))
            assert got == want

        def bad_or_good(j):
            if j == 7:
                try:
                    merged.evaluate_circuits([circuits[j]], [params[j][:-1]])
                except ValueError:
                    return "refused"
                return "accepted"
            return merged.evaluate_circuits([circuits[j]], [params[j]])[0]

        mixed = list(pool.map(bad_or_good, range(P)))
        assert mixed[7] == "refused" and mixed[:7] == want[:7] and mixed[8:] == want[8:]
    # raw ABI: an unknown circuit id is refused with the library's message
    dev = ev.statevector_device
    out = C.c_double(0.0)
    assert dev._lib.qsv_eval_coalesced(dev._handle, 10**6, None, 0, 0.0, C.byref(out)) == _lib.QSV_E_ARG
    assert b"unknown circuit id" in dev._lib.qsv_last_error(dev._handle)
    # two-circuit calls still go through the Python merger and agree
    assert merged.evaluate_circuits(circuits[:2], params[:2]) == want[:2]


def test_nft_last_layer_search_on_the_device():
    """The reference's operator test with its own optimiser (NFT, maxfev = 40; test/minimum_eigensolvers/evqe/solver.py:28-36,
    test_evqe_operators.py:91-93: "sum of expectation values decreased") on the GPU evaluator, n = 10."""
    from queasars_amd.evqe import EVQEPopulation
    from queasars_amd.evqe.solver import NFT, EVQEMinimumEigensolver, EVQEMinimumEigensolverConfiguration

    n = 10
    op = helpers.random_ising_operator(n, seed=12)
    ev = OperatorCircuitEvaluator(op)
    population = EVQEPopulation.random_population(n, 2, 8, False, 0)
    cfg = EVQEMinimumEigensolverConfiguration(
        optimizer=NFT(maxfev=40), population_size=8, max_generations=1, random_seed=0, n_initial_layers=2,
        randomize_initial_population_parameters=False, speciation_genetic_distance_threshold=2, use_tournament_selection=True,
        tournament_size=2, selection_alpha_penalty=0.1, selection_beta_penalty=0.1, parameter_search_probability=0.3,
        topological_search_probability=0.4, layer_removal_probability=0.05,
    )
    solver = EVQEMinimumEigensolver(cfg)

    def total(pop):
        cs = [i.get_parameterized_quantum_circuit() for i in pop.individuals]
        return sum(ev.evaluate_circuits(cs, [list(i.parameter_values) for i in pop.individuals]))

    before = total(population)
    searched, nfev = solver._last_layer_search(ev, population)
    after = total(searched)
    assert after < before and 8 * 40 <= nfev <= 8 * 42
    # every individual's value is what the oracle says for its new parameters
    ind = searched.individuals[3]
    got = ev.evaluate_circuits([ind.get_parameterized_quantum_circuit()], [list(ind.parameter_values)])[0]
    assert abs(got - helpers.oracle_expectation(ind.get_parameterized_quantum_circuit(), list(ind.parameter_values), op)) < EXP_TOL


# ---- (k) register splitting (csrc/split.hpp): the contraction path against the pass path and the oracle ----------------


def _split_device(n, split, **kwargs):
    """A device with register splitting on or off (the library reads QSV_SPLIT when the handle is created)."""
    import os

    old = os.environ.get("QSV_SPLIT")
    os.environ["QSV_SPLIT"] = "1" if split else "0"
    try:
        return StatevectorDevice(n, **kwargs)
    finally:
        if old is None:
            del os.environ["QSV_SPLIT"]
        else:
            os.environ["QSV_SPLIT"] = old


def _keys_like_the_library(circuit, n):
    """Number of keys of the split form the library finds (it tries virtual circuits of at most a tile, then a tile + 2 and
    + 4 qubits), -1 if there is none."""
    tile = 12 if n < 21 else 13
    for extra in (0, 2, 4):
        k = _split_keys(circuit, min(tile + extra, n - 1))
        if k >= 0:
            return k
    return -1


def _split_keys(circuit, max_side):
    ops = circuit.packed()
    cap = 4 * len(ops) + 64
    a, b = np.zeros(cap, dtype=QSV_OP_DTYPE), np.zeros(cap, dtype=QSV_OP_DTYPE)
    na, nb, mask = C.c_int(0), C.c_int(0), C.c_uint64(0)
    return _lib.load().qsv_split_describe(circuit.n_qubits, len(ops), _lib.as_ptr(ops), max_side, C.byref(mask), _lib.as_ptr(a), cap,
                                          C.byref(na), _lib.as_ptr(b), cap, C.byref(nb))


@pytest.mark.parametrize("n,layers,count", [(14, 5, 24), (16, 6, 32), (20, 4, 64), (20, 5, 32), (22, 5, 16)])
def test_split_evaluations_agree_with_the_pass_path(n, layers, count):
    """The same population through the split path (virtual circuits + contraction kernel) and through the ordinary
    multi-pass path: |dE| <= 1e-10, for circuits without a key, with one to three keys, with virtual circuits larger than
    a tile, and for circuits that cannot be split at all (deeper ones), mixed in one batch."""
    _, circuits, params = helpers.population_circuits(n, layers, count, seed=40 + n)
    op = helpers.random_ising_operator(n, seed=n)
    keys = [_keys_like_the_library(c, n) for c in circuits]
    # the population exercises keys, and key-less or unsplit circuits (or, since four and five keys are taken, those)
    assert max(keys) >= 1 and (min(keys) <= 0 or max(keys) >= 4), keys
    split = OperatorCircuitEvaluator(op, statevector_device=_split_device(n, True)).evaluate_circuits(circuits, params)
    plain = OperatorCircuitEvaluator(op, statevector_device=_split_device(n, False)).evaluate_circuits(circuits, params)
    assert np.abs(np.asarray(split) - np.asarray(plain)).max() < EXP_TOL
    # a small sample against the NumPy oracle as well
    for i in (0, count // 2, count - 1):
        if n <= 16:
            assert abs(split[i] - helpers.oracle_expectation(circuits[i], params[i], op)) < EXP_TOL


@pytest.mark.parametrize("n,layers,count", [(20, 6, 32), (20, 5, 48), (24, 6, 12)])
def test_split_evaluations_with_four_and_five_keys(n, layers, count, c_oracle):
    """Deeper individuals split with four and five cut keys (16 / 32 product terms: launch_factor_big, quadratic operators
    only): against the C oracle (1e-10), against the multi-pass path, bitwise the same alone as in the batch; under a
    general operator and in the sampler branch the same circuits take their ordinary plans."""
    _, circuits, params = helpers.population_circuits(n, layers, count, seed=0)
    limit = 16 if n < 21 else 17
    keys = [_split_keys(c, limit) for c in circuits]
    assert max(keys) >= 4, keys
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    ev.statevector_device.set_option("split_max_keys", 5)  # (the default since the chain stream; stated for the record)
    got = ev.evaluate_circuits(circuits, params)
    prof_dev = ev.statevector_device
    prof_dev.set_profiling(True)
    ev.evaluate_circuits(circuits, params)
    prof = prof_dev.profile()
    prof_dev.set_profiling(False)
    assert prof["kernel_states"][2] >= sum(1 for k in keys if k >= 4), "the circuits with four and five keys did not run split"
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    big = [i for i, k in enumerate(keys) if k >= 4]
    for i in (big[:6] if n < 24 else big[:2]) + [0]:
        assert abs(got[i] - c_oracle.evaluate(circuits[i], params[i], op, table, scratch)) < EXP_TOL, (i, keys[i])
    plain = OperatorCircuitEvaluator(op, statevector_device=_split_device(n, False)).evaluate_circuits(circuits, params)
    assert np.abs(np.asarray(got) - np.asarray(plain)).max() < EXP_TOL
    assert [ev.evaluate_circuits([circuits[i]], [params[i]])[0] for i in big[:4]] == [got[i] for i in big[:4]]
    if n == 20 and layers == 6:
        general = helpers.random_pauli_operator(n, 6, seed=4)
        sub_c, sub_p = [circuits[i] for i in big[:3]], [params[i] for i in big[:3]]
        # (the same circuit objects, registered with five keys on `ev`'s device: another operator on that device)
        a = OperatorCircuitEvaluator(general, statevector_device=ev.statevector_device).evaluate_circuits(sub_c, sub_p)
        b = OperatorCircuitEvaluator(general, statevector_device=_split_device(n, False)).evaluate_circuits(sub_c, sub_p)
        assert np.abs(np.asarray(a) - np.asarray(b)).max() < EXP_TOL
        sampler = OperatorSamplerCircuitEvaluator(20000, op, alpha=1.0, seed=3, statevector_device=ev.statevector_device)
        means = sampler.evaluate_circuits(sub_c, sub_p)
        spread = float(np.abs(op.coeffs).sum())
        assert np.abs(np.asarray(means) - np.asarray([got[i] for i in big[:3]])).max() < 0.05 * spread


@pytest.mark.parametrize("n,layers,count", [(8, 3, 6), (14, 4, 6), (20, 4, 4), (20, 7, 2), (12, 2, 4)])
def test_exact_probability_cvar_against_the_oracle(n, layers, count, c_oracle):
    """The sampler branch without sampling noise (``sampler_shots=None``): CVaR_alpha of the exact distribution on the device
    against the oracle's restatement of the reference's accumulation loop (oracle.cvar_expectation, expectation_calculation.py
    :14-32) fed with the oracle's own exact probabilities and values -- 1e-10, for alpha = 1, 0.5, 0.05 and one awkward
    value; split circuits (side tables), unsplittable ones (probabilities of the last pass), operators with many equal
    values (ties), one-tile registers."""
    from oracle import statevector_oracle as so

    _, circuits, params = helpers.population_circuits(n, layers, count, seed=11 + n)
    operators = [helpers.random_ising_operator(n, seed=5 + n)]
    if n <= 14:  # an operator with few distinct values: ties everywhere
        operators.append(PauliOperator.from_sparse_list([("Z", [0], 1.0), ("Z", [1], 1.0), ("ZZ", [2, 3], 2.0), ("Z", [n - 1], -1.0)], n))
    for op in operators:
        table = c_oracle.diagonal_table(op)
        for alpha in (1.0, 0.5, 0.05, 0.3217):
            ev = OperatorSamplerCircuitEvaluator(None, op, alpha=alpha)
            got = ev.evaluate_circuits(circuits, params)
            assert got == ev.evaluate_circuits(circuits, params)  # deterministic
            for i in range(count if n < 20 else 2):
                probs = np.abs(c_oracle.simulate(circuits[i], params[i])) ** 2
                if np.isclose(alpha, 1):
                    want = float(np.dot(probs, table))
                else:
                    want = so.cvar_expectation(list(zip(range(1 << n), probs.tolist(), table.tolist())), alpha)
                assert abs(got[i] - want) < EXP_TOL, (n, alpha, i, got[i], want)
            ev.statevector_device.close()
    with pytest.raises(ValueError):
        OperatorSamplerCircuitEvaluator(0, operators[0])


def test_config2_exact_workload_against_the_c_oracle(c_oracle):
    """BASELINE.json configs[1] exactly as bench.py runs it -- n = 20, P = 64, L = 4, population seed 0, 190 ZZ + 20 Z Ising
    operator of default_rng(2020) --: all 64 individuals against the C oracle (1e-10)."""
    n = 20
    _, circuits, params = helpers.population_circuits(n, 4, 64, seed=0)
    op = helpers.random_ising_operator(n, seed=2020)
    assert len(op) == 210
    got = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    ref = np.asarray([c_oracle.evaluate(c, p, op, table, scratch) for c, p in zip(circuits, params)])
    assert np.abs(got - ref).max() < EXP_TOL


def test_population_fixture_in_the_reference_wire_format_evaluates_to_its_stored_values():
    """tests/golden/population_n6.json (the reference's JSON layout, serialization.py:27-65): load -> evaluate on the device
    -> the stored expectation values (1e-10)."""
    from queasars_amd.evqe.serialization import population_from_dict

    data = json.loads((Path(__file__).resolve().parent / "golden" / "population_n6.json").read_text())
    population = population_from_dict(data["population"])
    op = PauliOperator(data["operator"]["labels"], data["operator"]["coeffs"])
    circuits = [ind.get_parameterized_quantum_circuit() for ind in population.individuals]
    params = [list(ind.parameter_values) for ind in population.individuals]
    got = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
    assert np.abs(got - np.asarray(data["expectations"])).max() < EXP_TOL


def test_split_results_do_not_depend_on_the_batch(c_oracle):
    """Bitwise: an evaluation's value is the same alone, in a batch of split evaluations and in a batch mixed with
    circuits that keep the ordinary plan; and it matches the C oracle at n = 20."""
    n = 20
    _, shallow, ps = helpers.population_circuits(n, 4, 12, seed=7)
    _, deep, pd = helpers.population_circuits(n, 9, 3, seed=8)
    assert all(_keys_like_the_library(c, n) < 0 for c in deep)
    op = helpers.random_ising_operator(n, seed=3)
    ev = OperatorCircuitEvaluator(op)
    alone = [ev.evaluate_circuits([c], [p])[0] for c, p in zip(shallow, ps)]
    together = ev.evaluate_circuits(shallow, ps)
    mixed = ev.evaluate_circuits(deep[:2] + shallow + deep[2:], pd[:2] + ps + pd[2:])
    assert together == alone and mixed[2:-1] == alone
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    ref = [c_oracle.evaluate(c, p, op, table, scratch) for c, p in zip(shallow[:4], ps[:4])]
    assert np.abs(np.asarray(alone[:4]) - np.asarray(ref)).max() < EXP_TOL


def test_one_launch_route_of_split_evaluations(c_oracle):
    """Split evaluations under a quadratic operator at n = 20: the launch that runs the two virtual circuits also forms
    their Gram matrices and combines them (kModeFusedFactor; the two sides' workgroups hand over through global memory).
    Same values as the three-launch route (another summation order: 1e-10), as the C oracle, in fp32 within the fp32
    bound -- and the SAME BITS on every one of many repetitions, alone or in company, on one stream or two: a hand-off
    that read a stale line would show as a repetition that differs."""
    n = 20
    _, circuits, params = helpers.population_circuits(n, 4, 64, seed=0)
    keys = [_keys_like_the_library(c, n) for c in circuits]
    assert max(keys) >= 2 and min(keys) == 0
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    one = ev.evaluate_circuits(circuits, params)
    dev.set_option("fused_factor", 0)
    three = ev.evaluate_circuits(circuits, params)
    dev.set_option("fused_factor", 1)
    assert np.abs(np.asarray(one) - np.asarray(three)).max() < EXP_TOL
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    for i in (0, 17, 63, int(np.argmax(keys))):
        assert abs(one[i] - c_oracle.evaluate(circuits[i], params[i], op, table, scratch)) < EXP_TOL
    for rep in range(400):
        assert ev.evaluate_circuits(circuits, params) == one, rep
    assert ev.evaluate_circuits(circuits[::-1], params[::-1]) == one[::-1]
    assert [ev.evaluate_circuits([c], [p])[0] for c, p in zip(circuits[:16], params[:16])] == one[:16]
    # more evaluations than side-table slots of one stream (several launch groups per push), twice over
    big_c, big_p = circuits * 5, params * 5
    for _ in range(2):
        assert ev.evaluate_circuits(big_c, big_p) == one * 5
    dev.set_option("streams", 1)
    assert ev.evaluate_circuits(circuits, params) == one
    # the sides' states handed over in LDS (small sides as they would lie in their slots, three-key sides of thirteen qubits as
    # padded rows read in place) or through their slots: the same sums in the same order
    dev.set_option("fused_lds_table", 0)
    assert ev.evaluate_circuits(circuits, params) == one
    dev.set_option("fused_lds_table", 1)
    # a side's values of D from a table of its own (filled when the plan is uploaded) or gathered from D: the same values
    dev.set_option("side_diag", 0)
    assert ev.evaluate_circuits(circuits, params) == one
    dev.set_option("side_diag", 1)
    assert ev.evaluate_circuits(circuits, params) == one
    # ... and they follow the operator: the same circuits, registered already, under another one and back
    other = helpers.random_ising_operator(n, seed=7)
    fresh = OperatorCircuitEvaluator(other).evaluate_circuits(circuits, params)
    assert max(abs(a - b) for a, b in zip(fresh, one)) > 1e-3
    shared = OperatorCircuitEvaluator(other, statevector_device=dev)  # (evaluators that share a device set their operator per call)
    assert shared.evaluate_circuits(circuits, params) == fresh
    assert ev.evaluate_circuits(circuits, params) == one
    assert shared.evaluate_circuits(circuits, params) == fresh
    got32 = np.asarray(OperatorCircuitEvaluator(op, dtype="fp32").evaluate_circuits(circuits, params))
    assert np.abs(got32 - np.asarray(one)).max() < FP32_REL * float(np.abs(op.coeffs).sum())


@pytest.mark.parametrize("layers", [4, 5])
def test_sides_of_eight_amplitudes_per_thread_and_half_sides(c_oracle, layers):
    """The one-launch route at 20 qubits plans its sides with eight amplitudes per thread (option "sides_r3", the default):
    sides of up to twelve virtual qubits as one tile, thirteen with one or two keys as two tiles swept by the side's one
    workgroup, three-key sides of thirteen as TWO workgroups each that trade half rows through memory (kEvalHalves) -- the
    benchmark's population has one of those, the five-layer one eighteen.  Against the same circuits on sixteen amplitudes
    per thread and one workgroup per side (other orders of the sums: 1e-10), against the C oracle for a circuit of every
    form, and the same bits on every repetition, alone or in company, in any order: a half side that read its partner's rows
    too early, or a stale line of them, would show as a repetition that differs."""
    n = 20
    _, circuits, params = helpers.population_circuits(n, layers, 64, seed=0)
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    got = ev.evaluate_circuits(circuits, params)
    wide = OperatorCircuitEvaluator(op)
    wide.statevector_device.set_option("sides_r3", 0)  # (applies to circuits registered afterwards: none yet)
    want = wide.evaluate_circuits(circuits, params)
    assert np.abs(np.asarray(got) - np.asarray(want)).max() < EXP_TOL
    assert got != want  # (the two forms add in different orders: were they the same bits, the option would not be doing anything)
    keys = [_keys_like_the_library(c, n) for c in circuits]
    assert 3 in keys
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    picked = sorted({keys.index(k) for k in set(keys) if k >= 0} | {0, 63})
    for i in picked:
        assert abs(got[i] - c_oracle.evaluate(circuits[i], params[i], op, table, scratch)) < EXP_TOL, (i, keys[i])
    for rep in range(200):
        assert ev.evaluate_circuits(circuits, params) == got, rep
    assert ev.evaluate_circuits(circuits[::-1], params[::-1]) == got[::-1]
    three = [i for i, k in enumerate(keys) if k == 3][:4]
    assert [ev.evaluate_circuits([circuits[i]], [params[i]])[0] for i in three] == [got[i] for i in three]
    assert ev.evaluate_circuits(circuits * 3, params * 3) == got * 3  # (more half sides than one launch takes: several launches)
    if layers == 4:
        # a push too large for one launch with the sides' states in LDS (128 evaluations, two of them with half sides): the half-sided
        # ones get a launch of their own beside the others' -- the same bits as in pushes of 64
        _, many_c, many_p = helpers.population_circuits(n, layers, 128, seed=0)
        many_keys = [_keys_like_the_library(c, n) for c in many_c]
        assert many_keys.count(3) >= 1 and len(many_c) == 128
        together = ev.evaluate_circuits(many_c, many_p)
        assert together == ev.evaluate_circuits(many_c[:64], many_p[:64]) + ev.evaluate_circuits(many_c[64:], many_p[64:])
        for rep in range(50):
            assert ev.evaluate_circuits(many_c, many_p) == together, rep
    # the form a thirteen-qubit side falls back to when its plan keeps the last key qubit inside the tile: two tiles swept by the
    # side's one workgroup (three keys: one tile of sixteen amplitudes per thread, as before)
    os.environ["QSV_NO_HALF_SIDES"] = "1"
    try:
        swept = OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params)
    finally:
        del os.environ["QSV_NO_HALF_SIDES"]
    assert np.abs(np.asarray(swept) - np.asarray(want)).max() < EXP_TOL
    # the same plans -- sides of eight amplitudes per thread, two tiles for a thirteen-qubit side -- where the route is not the
    # one-launch one: a general operator (the sides' states go to their tables, the term kernel reads them) and sampled
    # distributions of the same circuits (the split sampler), against a device that does not split
    general = helpers.random_pauli_operator(n, 6, seed=4)
    few = sorted({keys.index(k) for k in set(keys) if k >= 0})
    sub_c, sub_p = [circuits[i] for i in few], [params[i] for i in few]
    split = OperatorCircuitEvaluator(general, statevector_device=ev.statevector_device).evaluate_circuits(sub_c, sub_p)
    plain = OperatorCircuitEvaluator(general)
    plain.statevector_device.set_option("split", 0)
    assert np.abs(np.asarray(split) - np.asarray(plain.evaluate_circuits(sub_c, sub_p))).max() < EXP_TOL


def test_sides_tables_of_d_survive_their_buffer_filling_up():
    """The sides' own tables of D (kSplitSideDiag) live in a buffer that is emptied and filled again when it is full (every
    plan is then uploaded anew, with new tables): a population evaluated before, between and after enough other structures
    to fill it several times over gives the same bits every time, and the others agree with an evaluator of their own."""
    n = 20
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    _, first_c, first_p = helpers.population_circuits(n, 4, 64, seed=0)
    first = ev.evaluate_circuits(first_c, first_p)
    for seed in range(1, 13):  # (64 structures x 2 tables of up to 2^13 doubles each: the 4 MiB the buffer starts with hold four such populations)
        _, circuits, params = helpers.population_circuits(n, 3 + seed % 3, 64, seed=seed)
        got = ev.evaluate_circuits(circuits, params)
        if seed % 4 == 0:
            assert got == OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params)
            assert ev.evaluate_circuits(first_c, first_p) == first
    assert ev.evaluate_circuits(first_c, first_p) == first


def test_split_fp32_and_general_operators():
    """fp32 tables through the split path (within the fp32 bound of the fp64 value); a general operator on split circuits
    (the term kernel) against the same device with splitting off."""
    n = 18
    _, circuits, params = helpers.population_circuits(n, 4, 16, seed=5)
    op = helpers.random_ising_operator(n, seed=9)
    want = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
    got32 = np.asarray(OperatorCircuitEvaluator(op, dtype="fp32").evaluate_circuits(circuits, params))
    assert np.abs(got32 - want).max() < FP32_REL * float(np.abs(op.coeffs).sum())
    general = helpers.random_pauli_operator(n, 12, seed=4)
    a = OperatorCircuitEvaluator(general, statevector_device=_split_device(n, True)).evaluate_circuits(circuits[:4], params[:4])
    b = OperatorCircuitEvaluator(general, statevector_device=_split_device(n, False)).evaluate_circuits(circuits[:4], params[:4])
    assert np.abs(np.asarray(a) - np.asarray(b)).max() < EXP_TOL


def test_split_circuits_get_their_ordinary_plan_on_first_need():
    """A circuit registered in split form has no multi-pass plan until something needs its state: the read-out and the
    old sampler build it then (several at once on the worker threads), and the split evaluation still works afterwards."""
    n = 16
    _, circuits, params = helpers.population_circuits(n, 4, 8, seed=11)
    op = helpers.random_ising_operator(n, seed=2)
    ev = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    before = ev.evaluate_circuits(circuits, params)
    ref = [helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)]
    assert np.abs(np.asarray(before) - np.asarray(ref)).max() < EXP_TOL
    state = dev.statevector(circuits[0], params[0])
    assert np.abs(state - helpers.oracle_state(circuits[0], params[0])).max() < 1e-12
    states, _ = dev.sample_batch(circuits, params, shots=64, seed=5)  # (split sampler: needs no plan)
    assert np.asarray(states).shape == (8, 64)
    old = _sampler_device(n, False)
    old.set_operator(op)
    old.expectation_values(circuits, params)  # (registers the circuits in split form)
    states_old, _ = old.sample_batch(circuits, params, shots=64, seed=5)  # the eight plans in one go
    assert np.asarray(states_old).shape == (8, 64)
    probs = np.abs(helpers.oracle_state(circuits[3], params[3])) ** 2
    assert all(probs[int(s)] > 0 for s in np.asarray(states)[3])
    assert ev.evaluate_circuits(circuits, params) == before


@pytest.mark.parametrize("n", [10, 14])
def test_prepare_without_lds_staging(n):
    """prepare_eval stages the parameter vector and the folded gates' matrices in LDS when they fit (1024 parameters, 128
    folded gates); beyond that it reads and computes them one by one.  Both limits exceeded, in the prepare kernel
    (n = 10) and in the pass kernel's own preparation of a split evaluation's virtual circuits (n = 14)."""
    rng = np.random.default_rng(n)
    c = CircuitIR(n)
    k = 0
    from queasars_amd.ir import ParamRef

    def angle():
        nonlocal k
        k += 1
        return ParamRef(k - 1)

    for rep in range(20 if n == 10 else 26):  # u gates before anything entangles: 200 / 364 folded gates, 3 parameters each
        for q in range(n):
            c.u(angle(), angle(), angle(), q)
    for q in range(0, n - 1, 2):
        c.cu3(angle(), angle(), angle(), q, q + 1)
    for q in range(n):
        c.u(angle(), angle(), angle(), q)
    assert c.num_parameters == k and (k > 1024 or n == 10)
    params = list(rng.uniform(-np.pi, np.pi, size=k))
    op = helpers.random_ising_operator(n, seed=1)
    got = OperatorCircuitEvaluator(op).evaluate_circuits([c, c], [params, params[::-1]])
    ref = [helpers.oracle_expectation(c, p, op) for p in (params, params[::-1])]
    assert np.abs(np.asarray(got) - np.asarray(ref)).max() < EXP_TOL


@pytest.mark.parametrize("shots,alpha", [(512, 0.5), (500, 0.3), (1000, 1.0), (4096, 0.05), (37, 0.25)])
def test_cvar_on_the_device_equals_the_host_cvar_of_the_same_samples(shots, alpha):
    """qsv_sample_cvar_batch sorts the sample values on the device; the same seed through qsv_sample_batch gives the
    samples, whose CVaR the host computes the reference's way (expectation_calculation.py:27-40 for equal weights)."""
    from queasars_amd.circuit_evaluation.circuit_evaluation import _cvar_of_sample_matrix

    n = 11
    _, circuits, params = helpers.population_circuits(n, 3, 7, seed=21)
    op = helpers.random_ising_operator(n, seed=8)
    dev = StatevectorDevice(n)
    dev.set_operator(op)
    got = dev.sample_cvar_batch(circuits, params, shots, 99, alpha)
    _, values = dev.sample_batch(circuits, params, shots, 99, with_values=True)
    want = _cvar_of_sample_matrix(values, alpha)
    assert np.abs(np.asarray(got) - np.asarray(want)).max() < 1e-12 * max(1.0, float(np.abs(values).max()))
    with pytest.raises(ValueError):
        dev.sample_cvar_batch(circuits, params, 5000, 1, 0.5)


# ---- (l) sampling split circuits from their side tables (kernels.hpp: launch_split_sample) ------------------------------


def _sampler_device(n, split_sample, **kwargs):
    """A device that samples split circuits from their side tables (or, off, from the 2^n probabilities as everybody else)."""
    import os

    old = os.environ.get("QSV_SPLIT_SAMPLE")
    os.environ["QSV_SPLIT_SAMPLE"] = "1" if split_sample else "0"
    try:
        return StatevectorDevice(n, **kwargs)
    finally:
        if old is None:
            del os.environ["QSV_SPLIT_SAMPLE"]
        else:
            os.environ["QSV_SPLIT_SAMPLE"] = old


def _one_circuit_per_key_count(n, layers, count, seed, tile=12):
    _, circuits, params = helpers.population_circuits(n, layers, count, seed=seed)
    chosen = {}
    for c, p in zip(circuits, params):
        k = _keys_like_the_library(c, n)
        if k >= 0:  # (unsplittable circuits are the old sampler's, tested elsewhere)
            chosen.setdefault(k, (c, p))
    return chosen


@pytest.mark.parametrize("n,layers,seed", [(14, 5, 54), (16, 6, 56), (17, 4, 3)])
def test_split_sampler_follows_the_exact_distribution(n, layers, seed):
    """Per basis state: counts of 400 000 shots within 6.5 sigma of shots * p (p = |oracle amplitude|^2) and a chi-square
    over the states expected at least five times, Poisson bounds for the rarer ones, for one circuit of every number of
    keys the population offers; no state of probability zero is ever drawn; values are D[state]; the draw is a
    function of the seed."""
    chosen = _one_circuit_per_key_count(n, layers, 48, seed)
    assert len(chosen) >= 2 and max(chosen) >= 1, sorted(chosen)
    op = helpers.random_ising_operator(n, seed=n)
    dev = _sampler_device(n, True)
    dev.set_operator(op)
    circuits = [c for c, _ in chosen.values()]
    params = [p for _, p in chosen.values()]
    shots = 400_000
    states, values = dev.sample_batch(circuits, params, shots, seed=17, with_values=True)
    again, _ = dev.sample_batch(circuits, params, shots, seed=17)
    other, _ = dev.sample_batch(circuits, params, shots, seed=18)
    assert np.array_equal(states, again) and not np.array_equal(states, other)
    index = np.arange(1 << n, dtype=np.uint64)
    table = np.zeros(1 << n)
    for z, c in zip(op.z_mask, op.coeffs):  # D[i] = sum_k c_k (-1)^popcount(i & z_k)
        parity = index & z
        for shift in (32, 16, 8, 4, 2, 1):
            parity ^= parity >> np.uint64(shift)
        table += np.where(parity & np.uint64(1), -c.real, c.real)
    for i, (c, p) in enumerate(zip(circuits, params)):
        probs = np.abs(helpers.oracle_state(c, p)) ** 2
        counts = np.bincount(states[i].astype(np.int64), minlength=1 << n)
        assert counts[probs == 0.0].sum() == 0
        mean = shots * probs
        big = mean >= 5.0
        z = (counts[big] - mean[big]) / np.sqrt(mean[big] * (1.0 - probs[big]))
        assert np.abs(z).max() < 6.5, (n, i)
        dof = int(big.sum())
        assert abs(float((z * z).sum()) / dof - 1.0) < 6.0 * np.sqrt(2.0 / dof), (n, i)  # chi-square over those bins
        # rare states: Poisson tails per bin, and their total count
        assert (counts[~big] <= mean[~big] + 6.5 * np.sqrt(mean[~big]) + 8.0).all(), (n, i)
        rare = float(mean[~big].sum())
        assert abs(float(counts[~big].sum()) - rare) <= 6.0 * np.sqrt(rare) + 1.0, (n, i)
        assert np.abs(values[i] - table[states[i].astype(np.int64)]).max() < 1e-12 * float(np.abs(op.coeffs).sum())


def test_split_sampler_in_a_mixed_batch_and_against_the_old_sampler():
    """n = 20: split circuits and deep (unsplittable) ones in one call -- every evaluation's sample mean of D is within
    6 sigma of ITS exact expectation (the results are ordered by input index), with the split sampler and with the old
    one; the device-side CVaR equals the host's CVaR of the same samples."""
    from queasars_amd.circuit_evaluation.circuit_evaluation import _cvar_of_sample_matrix

    n, shots = 20, 4096
    _, shallow, ps = helpers.population_circuits(n, 4, 20, seed=7)
    _, deep, pd = helpers.population_circuits(n, 9, 3, seed=8)
    circuits = deep[:2] + shallow + deep[2:]
    params = pd[:2] + ps + pd[2:]
    op = helpers.random_ising_operator(n, seed=3)
    exact = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
    for split_sample in (True, False):
        dev = _sampler_device(n, split_sample)
        dev.set_operator(op)
        _, values = dev.sample_batch(circuits, params, shots, seed=5, with_values=True)
        mean = values.mean(axis=1)
        sigma = values.std(axis=1) / np.sqrt(shots) + 1e-12
        assert (np.abs(mean - exact) <= 6.0 * sigma).all(), (split_sample, np.abs(mean - exact) / sigma)
        got = dev.sample_cvar_batch(circuits, params, shots, 5, 0.25)
        want = _cvar_of_sample_matrix(values, 0.25)
        assert np.abs(np.asarray(got) - np.asarray(want)).max() < 1e-12 * max(1.0, float(np.abs(values).max()))


@pytest.mark.parametrize("n,dtype,count", [(18, "fp32", 8), (24, "fp64", 6), (28, "fp64", 2)])
def test_split_sampler_on_large_registers(n, dtype, count):
    """Sample means against the exact expectation where the 2^n probabilities are large (n = 24: 128 MiB, n = 28: 2 GiB
    per evaluation): the split sampler touches 2^(n/2)-sized tables only; fp32 side tables at n = 18."""
    shots = 4096
    _, circuits, params = helpers.population_circuits(n, 3 if n == 28 else 4, count, seed=n)
    assert any(_keys_like_the_library(c, n) >= 0 for c in circuits)
    op = helpers.random_ising_operator(n, seed=1)
    exact = np.asarray(OperatorCircuitEvaluator(op, dtype=dtype).evaluate_circuits(circuits, params))
    dev = _sampler_device(n, True, dtype=dtype)
    dev.set_operator(op)
    _, values = dev.sample_batch(circuits, params, shots, seed=11, with_values=True)
    mean = values.mean(axis=1)
    sigma = values.std(axis=1) / np.sqrt(shots) + 1e-9
    assert (np.abs(mean - exact) <= 6.0 * sigma + (1e-3 if dtype == "fp32" else 0.0)).all(), np.abs(mean - exact) / sigma


# ---- (m) split evaluations under a quadratic diagonal operator: no sweep over the 2^n indices (kernels.hpp: launch_factor) ---


def _factor_device(n, factor, **kwargs):
    """A device whose split evaluations use the factorised expectation (or, off, the contraction kernel)."""
    import os

    old = os.environ.get("QSV_FACTOR")
    os.environ["QSV_FACTOR"] = "1" if factor else "0"
    try:
        return StatevectorDevice(n, **kwargs)
    finally:
        if old is None:
            del os.environ["QSV_FACTOR"]
        else:
            os.environ["QSV_FACTOR"] = old


def _diagonal_operator(n, seed, kind):
    """quadratic: Ising terms, a constant, a repeated pair; fields: single-Z terms and a constant only;
    cubic: the Ising terms plus Z strings of weight 3 and 4 (not quadratic: the contraction kernel's case)."""
    rng = np.random.default_rng(seed)
    terms = [("", [], 0.75)]
    for i in range(n):
        terms.append(("Z", [i], float(rng.normal())))
    if kind != "fields":
        for i in range(n):
            for j in range(i + 1, n):
                if rng.random() < 0.6:
                    terms.append(("ZZ", [i, j], float(rng.normal())))
        terms.append(("ZZ", [0, n - 1], 0.5))  # (a pair that is there already, most likely: couplings add up)
        terms.append(("ZZ", [n - 1, 0], -0.25))
    if kind == "cubic":
        terms.append(("ZZZ", [0, n // 2, n - 1], 0.9))
        terms.append(("ZZZZ", [1, 2, n // 2 + 1, n - 2], -0.6))
    return PauliOperator.from_sparse_list(terms, n)


@pytest.mark.parametrize("n,layers,count", [(14, 5, 24), (17, 4, 24), (20, 4, 48), (22, 5, 16), (25, 3, 6)])
def test_factorised_expectation_agrees_with_the_contraction_and_the_pass_path(n, layers, count):
    """Quadratic diagonal operators (with a constant, single-Z terms, repeated pairs; or no couplings at all): the
    factorised expectation, the contraction kernel and the ordinary multi-pass path agree within 1e-10; an operator with
    terms of weight three and four is not quadratic and still agrees (it takes the contraction kernel either way)."""
    _, circuits, params = helpers.population_circuits(n, layers, count, seed=70 + n)
    keys = [_keys_like_the_library(c, n) for c in circuits]
    assert max(keys) >= (1 if n < 25 else 0), keys  # (n = 25: sides of 12 and 13 qubits, the largest tables)
    for kind in ("quadratic", "fields", "cubic"):
        op = _diagonal_operator(n, seed=n, kind=kind)
        factor = OperatorCircuitEvaluator(op, statevector_device=_factor_device(n, True)).evaluate_circuits(circuits, params)
        contract = OperatorCircuitEvaluator(op, statevector_device=_factor_device(n, False)).evaluate_circuits(circuits, params)
        scale = max(1.0, float(np.abs(op.coeffs).sum()) / 50.0)
        assert np.abs(np.asarray(factor) - np.asarray(contract)).max() < EXP_TOL * scale, kind
        if kind == "cubic":
            assert factor == contract  # (the same kernels ran)
        if n <= 22:
            plain = OperatorCircuitEvaluator(op, statevector_device=_split_device(n, False)).evaluate_circuits(circuits, params)
            assert np.abs(np.asarray(factor) - np.asarray(plain)).max() < EXP_TOL * scale, kind
        if n <= 14:
            for i in (0, count - 1):
                assert abs(factor[i] - helpers.oracle_expectation(circuits[i], params[i], op)) < EXP_TOL * scale


# ---- (n) results left on the device, and the chained all-gather of the sharded population ---------------------------------


def test_results_into_device_memory_without_waiting():
    """qsv_eval_set_output: the batch's results land in the caller's device buffer, qsv_eval_end(NULL) does not wait; they
    equal the waiting call's bit for bit (split and ordinary evaluations, diagonal and general operators), a waiting end
    with a device output also returns them on the host, and a call that follows an unfinished batch waits for it."""
    import torch

    n = 18
    _, shallow, ps = helpers.population_circuits(n, 4, 12, seed=3)
    _, deep, pd = helpers.population_circuits(n, 9, 3, seed=4)
    circuits, params = deep[:1] + shallow + deep[1:], pd[:1] + ps + pd[1:]
    for op in (helpers.random_ising_operator(n, seed=5), helpers.random_pauli_operator(n, 10, seed=6)):
        ev = OperatorCircuitEvaluator(op)
        dev = ev.statevector_device
        want = ev.evaluate_circuits(circuits, params)
        buf = torch.full((len(circuits),), float("nan"), dtype=torch.float64, device="cuda")
        for _ in range(3):  # (back to back: every call has to wait for the one before)
            assert ev.evaluate_circuits_to_device(circuits, params, buf.data_ptr())
        torch.cuda.synchronize()
        assert buf.cpu().tolist() == want
        assert ev.evaluate_circuits(circuits[::-1], params[::-1]) == want[::-1]  # right behind an unfinished batch
        # the streaming entry points by hand, with a waiting end
        lib, handle = dev._lib, dev._handle
        ids, _, _ = dev._batch_metadata(circuits)
        counts = np.asarray([len(p) for p in params], dtype=np.int64)
        packed = np.concatenate([np.asarray(p, dtype=np.float64) for p in params])
        buf.fill_(float("nan"))
        out = np.zeros(len(circuits))
        assert lib.qsv_eval_begin(handle, len(circuits), _lib.as_ptr(ids), _lib.as_ptr(counts)) == 0
        assert lib.qsv_eval_suggested_pushes(handle) in (1, 2)
        assert lib.qsv_eval_set_output(handle, C.c_void_p(buf.data_ptr())) == 0
        assert lib.qsv_eval_push(handle, 0, len(circuits), _lib.as_ptr(packed)) == 0
        assert lib.qsv_eval_set_output(handle, None) != 0  # (too late: after the first push)
        assert lib.qsv_eval_end(handle, _lib.as_ptr(out)) == 0
        assert out.tolist() == want and buf.cpu().tolist() == want
    # noisy estimators cannot leave their values on the device
    noisy = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=5), estimator_precision=0.1)
    assert not noisy.evaluate_circuits_to_device(circuits, params, buf.data_ptr())


def test_chained_all_gather_of_a_sharded_population():
    """evaluate_population_sharded over an RCCL group (of one rank: the box has one GPU) takes the chained path --
    evaluation into the collective's send buffer, all-gather and copy back on one stream -- and returns what the
    evaluator returns; QSV_GATHER_CHAIN=0 takes the staged path."""
    import os

    import torch
    import torch.distributed as dist

    from queasars_amd import distributed as qd

    n = 16
    _, circuits, params = helpers.population_circuits(n, 4, 24, seed=9)
    ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=1))
    want = ev.evaluate_circuits(circuits, params)
    import socket

    with socket.socket() as probe:  # (a port nobody holds: the group has one rank, nobody else needs to know it)
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        device = torch.device("cuda", 0)
        for _ in range(2):
            assert qd.evaluate_block_and_gather(ev, circuits, params, len(circuits), 1, 0, None, device) == want
        assert qd._gather(want, len(circuits), 1, 0, None, device) == want
        assert ev.evaluate_circuits(circuits, params) == want  # (the evaluator keeps working on the chain's stream)
        # the population's parameters as a matrix in device memory: the chain then touches the host only for the launches
        width = max(len(p) for p in params)
        host = np.zeros((len(params), width))
        for i, p in enumerate(params):
            host[i, : len(p)] = p
        matrix = torch.from_numpy(host).cuda()
        for _ in range(2):
            assert qd.evaluate_block_and_gather(ev, circuits, matrix, len(circuits), 1, 0, None, device) == want
        assert qd.evaluate_population_sharded(ev, circuits, matrix) == want
    finally:
        dist.destroy_process_group()


# ---- (o) split evaluations under a general Pauli operator: two small matrices per term (kernels.hpp: launch_factor_terms) ----


@pytest.mark.parametrize("n,layers,count,n_terms", [(14, 5, 16, 40), (17, 4, 16, 60), (20, 4, 24, 30), (24, 3, 4, 25)])
def test_general_operators_on_split_circuits(n, layers, count, n_terms):
    """Random Pauli strings over {I, X, Y, Z} (diagonal strings among them): the term kernel on the two side tables against
    the ordinary path (state + grouped expectation, QSV_FACTOR=0) within 1e-10, and against the NumPy oracle at n = 14;
    circuits that cannot be split ride along in the same batch."""
    _, circuits, params = helpers.population_circuits(n, layers, count, seed=90 + n)
    _, deep, pd = helpers.population_circuits(n, 9, 2, seed=91 + n) if n <= 20 else (None, [], [])
    circuits, params = deep[:1] + circuits + deep[1:], pd[:1] + params + pd[1:]
    keys = [_keys_like_the_library(c, n) for c in circuits]
    assert max(keys) >= 1 or n >= 24, keys
    op = helpers.random_pauli_operator(n, n_terms, seed=n)
    factor = OperatorCircuitEvaluator(op, statevector_device=_factor_device(n, True)).evaluate_circuits(circuits, params)
    plain = OperatorCircuitEvaluator(op, statevector_device=_factor_device(n, False)).evaluate_circuits(circuits, params)
    scale = max(1.0, float(np.abs(op.coeffs).sum()) / 20.0)
    assert np.abs(np.asarray(factor) - np.asarray(plain)).max() < EXP_TOL * scale
    if n <= 14:
        for i in (0, 1, len(circuits) - 1):
            assert abs(factor[i] - helpers.oracle_expectation(circuits[i], params[i], op)) < EXP_TOL * scale
    # fp32 side tables, within the fp32 bound
    if n == 17:
        got32 = OperatorCircuitEvaluator(op, dtype="fp32", statevector_device=_factor_device(n, True, dtype="fp32")).evaluate_circuits(circuits, params)
        assert np.abs(np.asarray(got32) - np.asarray(plain)).max() < FP32_REL * float(np.abs(op.coeffs).sum())


# ---- (p) registers of 26 and 28 qubits: virtual circuits of up to a tile + 4 qubits ------------------------------------


@pytest.mark.parametrize("n,count", [(26, 12), (28, 8)])
def test_split_evaluations_on_the_largest_registers(n, count):
    """At 26 and 28 qubits a split form needs virtual circuits larger than a tile + 2 qubits (13 + 13 or 14 + 14 own
    qubits plus keys): such circuits take the pass kernel several passes over up to 16 tiles.  Factorised expectation =
    contraction kernel (1e-10) for the whole population; the ordinary multi-pass path (a 1 / 4 GiB state per evaluation)
    for two of its circuits; sample means of the split sampler within 6 sigma."""
    _, circuits, params = helpers.population_circuits(n, 4, count, seed=n)
    keys = [_keys_like_the_library(c, n) for c in circuits]
    assert sum(k >= 0 for k in keys) >= count // 2 and max(keys) >= 1, keys
    op = helpers.random_ising_operator(n, seed=2)
    factor_dev = _factor_device(n, True)
    factor = OperatorCircuitEvaluator(op, statevector_device=factor_dev).evaluate_circuits(circuits, params)
    contract = OperatorCircuitEvaluator(op, statevector_device=_factor_device(n, False)).evaluate_circuits(circuits, params)
    scale = max(1.0, float(np.abs(op.coeffs).sum()) / 50.0)
    assert np.abs(np.asarray(factor) - np.asarray(contract)).max() < EXP_TOL * scale
    some = [i for i, k in enumerate(keys) if k >= 0][:2]
    plain = OperatorCircuitEvaluator(op, statevector_device=_split_device(n, False)).evaluate_circuits(
        [circuits[i] for i in some], [params[i] for i in some])
    assert np.abs(np.asarray(plain) - np.asarray([factor[i] for i in some])).max() < EXP_TOL * scale
    shots = 4096
    split_only = [i for i, k in enumerate(keys) if k >= 0]
    _, values = factor_dev.sample_batch([circuits[i] for i in split_only], [params[i] for i in split_only], shots, seed=3, with_values=True)
    exact = np.asarray([factor[i] for i in split_only])
    sigma = values.std(axis=1) / np.sqrt(shots) + 1e-9
    assert (np.abs(values.mean(axis=1) - exact) <= 6.0 * sigma).all()


# ---- round 3, second half: multiplexed gates, the chain stream, results watched in the pinned buffer ------------------


def _device_with_env(n, dtype="fp64", **env):
    """A device created under the given environment (the library reads its plan and path switches when a handle is made)."""
    import os

    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return StatevectorDevice(n, dtype)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("n,layers,count", [(14, 8, 12), (20, 8, 6), (22, 6, 4)])
def test_multiplexed_gates_agree_with_separate_gates(n, layers, count, c_oracle):
    """Deep individuals through the multi-pass path with the u gates multiplied into their neighbouring cu3 (plan.hpp
    FUSION: two schedule entries with complementary predicates, products with a complex m00) and without: the same
    expectation values to 1e-12, both within 1e-10 of the C oracle; the amplitudes themselves at n = 14."""
    _, circuits, params = helpers.population_circuits(n, layers, count, seed=7)
    op = helpers.random_ising_operator(n, seed=n)
    fused_dev, plain_dev = _device_with_env(n, QSV_FUSE=1, QSV_SPLIT=0), _device_with_env(n, QSV_FUSE=0, QSV_SPLIT=0)
    fused = OperatorCircuitEvaluator(op, statevector_device=fused_dev).evaluate_circuits(circuits, params)
    plain = OperatorCircuitEvaluator(op, statevector_device=plain_dev).evaluate_circuits(circuits, params)
    assert np.abs(np.asarray(fused) - np.asarray(plain)).max() < 1e-12
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    for i in (0, count - 1):
        want = c_oracle.evaluate(circuits[i], params[i], op, table, scratch)
        assert abs(fused[i] - want) < EXP_TOL
    if n <= 14:
        for c, p in zip(circuits[:3], params[:3]):
            a, b = fused_dev.statevector(c, p), plain_dev.statevector(c, p)
            assert np.abs(a - b).max() < 1e-12
            assert np.abs(a - helpers.oracle_state(c, p)).max() < 1e-12


def test_multiplexed_gates_in_fp32():
    """fp32 plans carry multiplexed gates like fp64 plans since round 4 (the generated fp32 round loop has one packed-arithmetic
    body for u-type matrices and products with a complex m00); QSV_FUSE=0 switches them off: the same values, to fp32."""
    n = 16
    _, circuits, params = helpers.population_circuits(n, 7, 8, seed=9)
    op = helpers.random_ising_operator(n, seed=n)
    want = np.asarray(OperatorCircuitEvaluator(op, statevector_device=_device_with_env(n, QSV_SPLIT=0)).evaluate_circuits(circuits, params))
    bound = FP32_REL * float(np.abs(op.coeffs).sum())
    for fuse in (0, 1):
        dev = _device_with_env(n, "fp32", QSV_FUSE=fuse, QSV_SPLIT=0)
        got = np.asarray(OperatorCircuitEvaluator(op, statevector_device=dev).evaluate_circuits(circuits, params))
        assert np.abs(got - want).max() < bound, fuse


@pytest.mark.parametrize("tile_bits,reg_bits", [(0, 0), (13, 4), (12, 4), (11, 3), (10, 2), (9, 1)])
def test_fp32_round_loop_geometries(tile_bits, reg_bits):
    """The generated fp32 round loop (RoundLoopF32: packed butterflies, lane swaps, LDS exchanges of whole elements) in every
    register width, on deep unsplit circuits whose plans have all three kinds of round: amplitudes within 2e-5 of the oracle,
    expectation values within FP32_REL * sum |c_k| of fp64, and the same bits whatever the batch."""
    n = 15
    _, circuits, params = helpers.population_circuits(n, 7, 6, seed=21)
    op = helpers.random_ising_operator(n, seed=n)
    want = np.asarray([helpers.oracle_expectation(c, p, op) for c, p in zip(circuits, params)])
    bound = FP32_REL * float(np.abs(op.coeffs).sum())
    env = dict(QSV_SPLIT=0)
    if tile_bits:
        env.update(QSV_TILE_BITS=tile_bits, QSV_REG_BITS=reg_bits)
    dev = _device_with_env(n, "fp32", **env)
    ev = OperatorCircuitEvaluator(op, statevector_device=dev)
    got = np.asarray(ev.evaluate_circuits(circuits, params))
    assert np.abs(got - want).max() < bound
    for c, p in zip(circuits[:2], params[:2]):
        assert np.abs(dev.statevector(c, p) - helpers.oracle_state(c, p)).max() < 2e-5
    assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits[::-1], params[::-1]))[::-1], got)
    assert ev.evaluate_circuits([circuits[3]], [params[3]])[0] == got[3]
    # a general operator: the state is stored and read back by the grouped expectation kernel
    general = helpers.random_pauli_operator(n, 10, seed=2)
    got = np.asarray(OperatorCircuitEvaluator(general, statevector_device=dev).evaluate_circuits(circuits, params))
    want = np.asarray([helpers.oracle_expectation(c, p, general) for c, p in zip(circuits, params)])
    assert np.abs(got - want).max() < FP32_REL * float(np.abs(general.coeffs).sum())


def test_chain_stream_and_result_polling_change_no_bit():
    """A five-layer population at 20 qubits holds split evaluations of both kinds (one launch / launches of their own, among
    them circuits with four and five keys) and unsplit ones: the chain stream (the second kind beside the first) and the
    end of the batch read off the result buffer are scheduling only -- every value the same bits with either switched off,
    in one push and in two, and over 100 repetitions (the last-arriving workgroup of the 32-term Gram matrices adds the
    slices in slice order whoever it is)."""
    n = 20
    _, circuits, params = helpers.population_circuits(n, 5, 64, seed=0)
    _, deeper, deeper_params = helpers.population_circuits(n, 6, 24, seed=0)
    circuits, params = circuits + deeper, params + deeper_params
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    base = np.asarray(ev.evaluate_circuits(circuits, params))
    assert np.isfinite(base).all()
    for name in ("chain_stream", "poll_results"):
        dev.set_option(name, 0)
        assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, params)), base), name
        dev.set_option(name, 1)
    for _ in range(100):
        assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, params)), base)
    # the first 64 alone (one push: the chain stream is in use), and one at a time
    assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits[:64], params[:64])), base[:64])
    for i in (0, 17, 63, 70, 87):
        assert ev.evaluate_circuits([circuits[i]], [params[i]])[0] == base[i]
    plain = OperatorCircuitEvaluator(op, statevector_device=_split_device(n, False)).evaluate_circuits(circuits[60:70], params[60:70])
    assert np.abs(np.asarray(plain) - base[60:70]).max() < EXP_TOL


def test_repeated_batches_keep_their_layout_and_nothing_else(c_oracle):
    """The previous batch again (same circuit ids and counts, nothing changed in between) reuses its layout in the staging
    buffer.  Whatever comes between two calls must either leave that layout valid or invalidate it: new parameter values,
    another batch, options, a registration, the same batch pushed in pieces -- every result against the C oracle, and
    bitwise what a handle that never reuses a layout returns."""
    n = 20
    _, circuits, params = helpers.population_circuits(n, 5, 40, seed=3)
    op = helpers.random_ising_operator(n, seed=2020)
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    ev = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    ref_ev = OperatorCircuitEvaluator(op)
    ref_ev.statevector_device.set_option("repeat_layout", 0)
    rng = np.random.default_rng(5)

    def check(cs, ps, oracle_at=()):
        got = np.asarray(ev.evaluate_circuits(cs, ps))
        want = np.asarray(ref_ev.evaluate_circuits(cs, ps))
        assert np.array_equal(got, want)
        for i in oracle_at:
            assert abs(got[i] - c_oracle.evaluate(cs[i], ps[i], op, table, scratch)) < EXP_TOL
        return got

    first = check(circuits, params, oracle_at=(0, 39))
    assert np.array_equal(check(circuits, params), first)                      # the same batch again
    shifted = [[v + rng.normal(0.0, 0.3) for v in p] for p in params]
    moved = check(circuits, shifted, oracle_at=(7,))                           # ... with other parameter values
    assert np.abs(moved - first).max() > 1e-3
    check(circuits[:13], params[:13], oracle_at=(12,))                         # another batch in between
    assert np.array_equal(check(circuits, params), first)
    dev.set_option("chain_stream", 0)                                          # an option in between
    assert np.array_equal(check(circuits, params), first)
    dev.set_option("chain_stream", 1)
    assert np.array_equal(check(circuits, params), first)
    _, extra, extra_params = helpers.population_circuits(n, 4, 3, seed=77)     # a registration in between
    check(extra, extra_params, oracle_at=(1,))
    assert np.array_equal(check(circuits, params), first)
    assert np.array_equal(check(list(reversed(circuits)), list(reversed(params))), first[::-1])
    # the same batch pushed in pieces right after it went through in one: the kept layout is given up
    assert np.array_equal(check(circuits, params), first)
    dev._push_evals = 16
    try:
        assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, params)), first)
        assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, shifted)), moved)
    finally:
        dev._push_evals = 0
    assert np.array_equal(check(circuits, shifted), moved)


def test_fp32_split_evaluations_with_four_and_five_keys():
    """The 16- and 32-term Gram kernels (eight slices, last workgroup adds them) and the chain stream with fp32 side tables:
    six-layer individuals at 20 qubits, most of them with four or five keys, within the fp32 bound of the fp64 values and
    the same bits when the batch is evaluated again and one circuit at a time."""
    n = 20
    _, circuits, params = helpers.population_circuits(n, 6, 24, seed=0)
    keys = [_split_keys(c, 16) for c in circuits]
    assert sum(1 for k in keys if k >= 4) >= 6, keys
    op = helpers.random_ising_operator(n, seed=2020)
    want = np.asarray(OperatorCircuitEvaluator(op).evaluate_circuits(circuits, params))
    ev32 = OperatorCircuitEvaluator(op, dtype="fp32")
    got = np.asarray(ev32.evaluate_circuits(circuits, params))
    assert np.abs(got - want).max() < FP32_REL * float(np.abs(op.coeffs).sum())
    assert np.array_equal(np.asarray(ev32.evaluate_circuits(circuits, params)), got)
    for i in (0, 11, 23):
        assert ev32.evaluate_circuits([circuits[i]], [params[i]])[0] == got[i]


@pytest.mark.gpu
@pytest.mark.parametrize("n,layers,count", [(12, 3, 20), (16, 5, 24), (20, 4, 64), (20, 6, 24), (22, 4, 12)])
def test_parameter_values_resident_in_device_memory(n, layers, count, c_oracle):
    """qsv_eval_push_device: the kernels read the parameter values from a matrix in device memory (one row per circuit, rows
    padded to the longest) instead of the pinned staging buffer.  Bitwise what the same values give as host lists, whatever
    route an evaluation takes (one tile, split, one launch, multi-pass, mixed), also with the matrix produced on the torch
    stream right before the call (the ready event), with rubbish in the padding, and again with other values in place."""
    import torch

    _, circuits, params = helpers.population_circuits(n, layers, count, seed=n + layers)
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    width = max(len(p) for p in params) + 3
    host = np.full((count, width), 1e300)  # (never read: a circuit takes the first num_parameters values of its row)
    for i, p in enumerate(params):
        host[i, : len(p)] = p
    want = np.asarray(ev.evaluate_circuits(circuits, params))
    matrix = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()
    got = np.asarray(ev.evaluate_circuits(circuits, matrix))
    assert np.array_equal(got, want)
    assert np.array_equal(ev.evaluate_device_parameters(circuits, matrix, ready=True), want)  # (the kept layout, no event)
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    for i in (0, count - 1):
        assert abs(got[i] - c_oracle.evaluate(circuits[i], params[i], op, table, scratch)) < EXP_TOL
    # other values, written into the same matrix by a kernel queued right before the call
    shifted = host.copy()
    for i, p in enumerate(params):
        shifted[i, : len(p)] += 0.25
    want2 = np.asarray(ev.evaluate_circuits(circuits, [row[: len(p)].tolist() for row, p in zip(shifted, params)]))
    staged = torch.from_numpy(shifted).pin_memory()
    big = torch.empty(64 << 20, dtype=torch.float64, device="cuda")
    big.fill_(1.0)                       # (something for the stream to be busy with)
    matrix.copy_(staged, non_blocking=True)
    got2 = np.asarray(ev.evaluate_circuits(circuits, matrix))
    assert np.array_equal(got2, want2)
    assert np.abs(got2 - got).max() > 1e-6
    # a part of the population, as a view of rows (contiguous), and the host path right after on the same handle
    assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits[3:11], matrix[3:11])), want2[3:11])
    # rows wider than the preparation's LDS staging of a parameter vector (1024 values): read straight from the matrix
    wide = torch.full((count, 1100), 7.0, dtype=torch.float64, device="cuda")
    wide[:, :width] = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()
    # (that path takes every angle's sine and cosine where it needs them instead of once per angle in LDS: the same values
    # to the last bits, not bit for bit)
    assert np.abs(np.asarray(ev.evaluate_circuits(circuits, wide)) - want).max() < 1e-12
    assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, params)), want)


@pytest.mark.gpu
def test_device_resident_parameters_argument_checks_and_mixed_pushes():
    import torch

    n, count = 16, 12
    _, circuits, params = helpers.population_circuits(n, 4, count, seed=9)
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    want = np.asarray(ev.evaluate_circuits(circuits, params))
    width = max(len(p) for p in params)
    host = np.zeros((count, width))
    for i, p in enumerate(params):
        host[i, : len(p)] = p
    matrix = torch.from_numpy(host).cuda()
    with pytest.raises(ValueError):
        ev.evaluate_circuits(circuits, matrix.float())
    with pytest.raises(ValueError):
        ev.evaluate_circuits(circuits[:-1], matrix)
    with pytest.raises(ValueError):
        ev.evaluate_circuits(circuits, matrix[:, : width - 1].contiguous())  # rows shorter than the longest circuit needs
    with pytest.raises(ValueError):
        ev.evaluate_circuits(circuits, matrix.t().contiguous().t())          # not row-major
    assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, matrix)), want)  # the handle is fine after the errors
    # raw C ABI: a host pointer is refused; host and device pushes in one batch
    lib, handle = dev._lib, dev._handle
    ids, _need, _ = dev._batch_metadata(circuits)
    counts = np.full(count, width, dtype=np.int64)
    out = np.zeros(count)
    assert lib.qsv_eval_begin(handle, count, _lib.as_ptr(ids), _lib.as_ptr(counts)) == 0
    assert lib.qsv_eval_push_device(handle, 0, 5, _lib.as_ptr(host), None) == _lib.QSV_E_ARG
    assert lib.qsv_eval_end(handle, _lib.as_ptr(out)) != 0  # (not every evaluation was pushed)
    torch.cuda.synchronize()
    assert lib.qsv_eval_begin(handle, count, _lib.as_ptr(ids), _lib.as_ptr(counts)) == 0
    first_rows = np.ascontiguousarray(host[:5])
    assert lib.qsv_eval_push(handle, 0, 5, _lib.as_ptr(first_rows)) == 0
    assert lib.qsv_eval_push_device(handle, 5, count - 5, C.c_void_p(matrix.data_ptr() + 5 * width * 8), None) == 0
    assert lib.qsv_eval_end(handle, _lib.as_ptr(out)) == 0
    assert np.array_equal(out, want)
    # results left on the device as well: nothing of the step touches the host but the launch
    result = torch.zeros(count, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert dev.expectation_values_of_device_parameters(circuits, matrix.data_ptr(), width, 0, result.data_ptr()) is None
    torch.cuda.synchronize()
    assert np.array_equal(result.cpu().numpy(), want)


@pytest.mark.gpu
def test_generations_of_fresh_structures_outgrow_the_plan_arena(c_oracle):
    """An EVQE run registers new structures every generation and drops old ones.  The device-side plan arena (4 MB at first)
    fills up, is rebuilt from the plans still in use and finally grows: survivors of early generations must keep evaluating
    to the same bits through every rebuild, new structures must agree with the C oracle, and the ids of collected circuits
    are destroyed on the way (the handle's circuit table does not grow without bound)."""
    import gc

    n = 16
    op = helpers.random_ising_operator(n, seed=2020)
    table = c_oracle.diagonal_table(op)
    scratch = np.zeros(2 << n)
    ev = OperatorCircuitEvaluator(op)
    dev = ev.statevector_device
    rng = np.random.default_rng(11)
    survivors, survivor_params, survivor_values = [], [], []
    registered_high_water = 0
    for generation in range(24):
        _, fresh, fresh_params = helpers.population_circuits(n, 3 + generation % 4, 72, seed=100 + generation)
        circuits, params = survivors + fresh, survivor_params + fresh_params
        feed = params if generation % 3 else [np.asarray(p, dtype=np.float64) for p in params]
        got = np.asarray(ev.evaluate_circuits(circuits, feed))
        assert np.array_equal(got[: len(survivors)], np.asarray(survivor_values)), f"generation {generation}: a survivor's value moved"
        for i in rng.choice(len(fresh), size=2, replace=False):
            j = len(survivors) + int(i)
            assert abs(got[j] - c_oracle.evaluate(circuits[j], params[j], op, table, scratch)) < EXP_TOL
        assert np.array_equal(np.asarray(ev.evaluate_circuits(circuits, feed)), got)  # (the kept layout, right after a rebuild too)
        # three of this generation live on, the rest is dropped (their plans stay in the arena until it is rebuilt)
        base = len(circuits) - len(fresh)
        for i in rng.choice(len(fresh), size=3, replace=False):
            survivors.append(fresh[int(i)])
            survivor_params.append(fresh_params[int(i)])
            survivor_values.append(got[base + int(i)])
        del fresh, circuits, feed
        gc.collect()
        registered_high_water = max(registered_high_water, len(dev._watched))
    # 24 generations of 72 structures were registered, at most a generation and the survivors are alive at any time
    assert registered_high_water <= 72 + 3 * 24 + 72
    final = np.asarray(ev.evaluate_circuits(survivors, survivor_params))
    assert np.array_equal(final, np.asarray(survivor_values))
    # one batch whose plans alone exceed the arena: it grows
    _, many, many_params = helpers.population_circuits(n, 5, 520, seed=999)
    got = np.asarray(ev.evaluate_circuits(survivors + many, survivor_params + many_params))
    assert np.array_equal(got[: len(survivors)], np.asarray(survivor_values))
    for j in (0, 519):
        assert abs(got[len(survivors) + j] - c_oracle.evaluate(many[j], many_params[j], op, table, scratch)) < EXP_TOL
    assert np.array_equal(np.asarray(ev.evaluate_circuits(survivors, survivor_params)), np.asarray(survivor_values))


@pytest.mark.gpu
@pytest.mark.parametrize("with_checker,with_initial_state", [(False, False), (True, False), (True, True)])
def test_spsa_search_with_its_state_on_the_device(with_checker, with_initial_state):
    """evqe/device_search.py: iterates, sign vectors, points and values in device memory, every iteration queued on one stream
    without the host waiting.  Against the host driver (which gives every run bit for bit its own SPSA's iterates): the same
    stopping iteration for every run, iterates to 1e-9 (the trust region's norm is summed in another order), with the
    reference's termination rule as array operations, and with an initial-state circuit in front of every circuit."""
    from queasars_amd.evqe import EVQEPopulation
    from queasars_amd.evqe import solver as S

    n = 14
    pop = EVQEPopulation.random_population(n, 3, 24, True, 5)
    initial = None
    if with_initial_state:
        initial = CircuitIR(n)
        for q in range(n):
            initial.u(q, 0.3 + 0.1 * q, 0.2, -0.4)
    ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020), initial_state_circuit=initial)
    checker = S.SPSATerminationChecker(0.02, 1) if with_checker else None
    cfg = S.SPSA(maxiter=20, termination_checker=checker)

    def jobs():
        return [(ind.get_partially_parameterized_quantum_circuit({-1}), cfg.new_run(ind.get_layer_parameter_values(-1), seed=k))
                for k, ind in enumerate(pop.individuals)]

    host = jobs()
    S._minimize_batched(ev, host)
    for _ in range(2):  # (twice: the second search finds the stream, the buffers and the layouts of the first)
        device = jobs()
        S._minimize_batched(ev, device, on_device=True)
        assert [run.iteration for _, run in device] == [run.iteration for _, run in host]
        assert [run.nfev for _, run in device] == [run.nfev for _, run in host]
        assert all(run.done for _, run in device)
        for (_, a), (_, b) in zip(device, host):
            assert np.abs(a.x - b.x).max() < 1e-9
    if with_checker:
        assert len({run.iteration for _, run in host}) > 1  # (the runs did stop at different iterations)
    # the same arithmetic as torch operations instead of the library's one launch per iteration
    import os

    os.environ["QSV_DEVICE_SEARCH_TORCH"] = "1"
    try:
        by_torch = jobs()
        S._minimize_batched(ev, by_torch, on_device=True)
    finally:
        del os.environ["QSV_DEVICE_SEARCH_TORCH"]
    assert [run.iteration for _, run in by_torch] == [run.iteration for _, run in host]
    for (_, a), (_, b) in zip(by_torch, device):
        assert np.abs(a.x - b.x).max() < 1e-12
    # the evaluator serves ordinary calls right after
    circuits = [ind.get_parameterized_quantum_circuit() for ind in pop.individuals[:4]]
    params = [list(ind.parameter_values) for ind in pop.individuals[:4]]
    again = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020), initial_state_circuit=initial)
    assert ev.evaluate_circuits(circuits, params) == again.evaluate_circuits(circuits, params)


@pytest.mark.gpu
def test_spsa_step_entry_point_by_hand():
    """qsv_spsa_step through raw ctypes: argument checks; a proposal is x +- eps * delta rounded as NumPy rounds it; an accepted
    step is NumPy's expression for runs that are active, nothing for those that are not; maxiter stops a run."""
    import torch

    dev = StatevectorDevice(8)
    lib, handle = dev._lib, dev._handle
    rng = np.random.default_rng(3)
    n_runs, width, eps, lr = 5, 37, 0.35, 0.43
    x0 = rng.normal(size=(n_runs, width))
    delta = 1.0 - 2.0 * rng.integers(0, 2, size=(2, n_runs, width))
    f = rng.normal(size=2 * n_runs)
    x = torch.from_numpy(x0.copy()).cuda()
    signs = torch.from_numpy(delta.copy()).cuda()
    values = torch.from_numpy(f.copy()).cuda()
    points = torch.zeros((2 * n_runs, width), dtype=torch.float64, device="cuda")
    active = torch.tensor([1, 1, 0, 1, 1], dtype=torch.uint8, device="cuda")
    iterations = torch.tensor([0, 0, 0, 2, 0], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()

    def args(**kw):
        a = _lib.QsvSpsaStepArgs(n_runs=n_runs, width=width, x=x.data_ptr(), active=active.data_ptr(), iterations=iterations.data_ptr(),
                                 delta_accept=None, values=None, delta_propose=None, points=points.data_ptr(), eps=eps, lr=lr,
                                 trust_region=0, maxiter=3, window=0, reserved=0, min_rel=0.0, maxfev=-1, previous=None, n_values=None,
                                 changes=None)
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    assert lib.qsv_spsa_step(handle, None) == _lib.QSV_E_ARG
    assert lib.qsv_spsa_step(handle, C.byref(args(x=None))) == _lib.QSV_E_ARG
    assert lib.qsv_spsa_step(handle, C.byref(args(values=values.data_ptr()))) == _lib.QSV_E_ARG       # no signs to go with them
    assert lib.qsv_spsa_step(handle, C.byref(args(delta_propose=signs.data_ptr(), points=None))) == _lib.QSV_E_ARG
    assert lib.qsv_spsa_step(handle, C.byref(args(window=2))) == _lib.QSV_E_ARG                       # a rule without its state
    assert lib.qsv_spsa_step(handle, C.byref(args(eps=0.0))) == _lib.QSV_E_ARG
    # propose only
    assert lib.qsv_spsa_step(handle, C.byref(args(delta_propose=signs[0].data_ptr()))) == 0
    torch.cuda.synchronize()
    got = points.cpu().numpy()
    assert np.array_equal(got[0::2], x0 + eps * delta[0]) and np.array_equal(got[1::2], x0 - eps * delta[0])
    # accept with signs[0], propose with signs[1]
    assert lib.qsv_spsa_step(handle, C.byref(args(delta_accept=signs[0].data_ptr(), values=values.data_ptr(),
                                                  delta_propose=signs[1].data_ptr()))) == 0
    torch.cuda.synchronize()
    update = ((f[0::2] - f[1::2]) / (2 * eps))[:, None] * delta[0]
    want = x0 - (update * lr) * np.array([1, 1, 0, 1, 1])[:, None]
    assert np.array_equal(x.cpu().numpy(), want)
    assert iterations.cpu().tolist() == [1, 1, 0, 3, 1] and active.cpu().tolist() == [1, 1, 0, 0, 1]  # (run 3 reached maxiter)
    got = points.cpu().numpy()
    assert np.array_equal(got[0::2], want + eps * delta[1]) and np.array_equal(got[1::2], want - eps * delta[1])
    # with the trust region: updates longer than 1 are divided by their norm (summed in the device's order: to the last bits)
    x.copy_(torch.from_numpy(x0))
    active.fill_(1)
    iterations.zero_()
    torch.cuda.synchronize()
    assert lib.qsv_spsa_step(handle, C.byref(args(delta_accept=signs[0].data_ptr(), values=values.data_ptr(), trust_region=1))) == 0
    torch.cuda.synchronize()
    norm = np.sqrt((update * update).sum(axis=1))
    assert (norm > 1).any() and (norm < 1).any()
    scaled = np.where(norm[:, None] > 1, update / norm[:, None], update) * lr
    assert np.abs(x.cpu().numpy() - (x0 - scaled)).max() < 1e-14
    dev.close()


@pytest.mark.gpu
def test_a_layer_searched_inside_the_fully_parameterised_circuit_gives_the_bound_circuits_values():
    """The EVQE driver evaluates a layer's points on the individual's fully parameterised circuit (shared per structure) with
    the other layers' values as parameter values; the reference binds those into the circuit.  Bit for bit the same
    expectation values on the device (the plan of a circuit does not depend on whether an angle is a literal), for split and
    unsplit individuals; and a whole EVQE run is the same run either way."""
    import os

    from queasars_amd.evqe import EVQEPopulation
    from queasars_amd.evqe import solver as S

    n = 16
    op = helpers.random_ising_operator(n, seed=2020)
    ev = OperatorCircuitEvaluator(op)
    rng = np.random.default_rng(2)
    for layers in (3, 7):
        pop = EVQEPopulation.random_population(n, layers, 10, True, layers)
        bound_circuits, bound_rows, full_circuits, full_rows = [], [], [], []
        for ind in pop.individuals:
            layer = int(rng.integers(0, layers))
            positions = list(ind.layer_parameter_indices[layer])
            point = rng.uniform(0, 6.28, size=len(positions))
            bound_circuits.append(ind.get_partially_parameterized_quantum_circuit({layer}))
            bound_rows.append(point.tolist())
            full = np.asarray(ind.parameter_values, dtype=np.float64)
            full[positions] = point
            full_circuits.append(ind.get_parameterized_quantum_circuit(shared=True))
            full_rows.append(full.tolist())
        assert ev.evaluate_circuits(bound_circuits, bound_rows) == ev.evaluate_circuits(full_circuits, full_rows)
    cfg = dict(optimizer=S.SPSA(maxiter=8), population_size=12, max_generations=3, random_seed=4, n_initial_layers=2,
               randomize_initial_population_parameters=True, use_tournament_selection=True, tournament_size=2,
               parameter_search_probability=0.5, topological_search_probability=0.6, device_resident_search=False)
    results = []
    for share in ("2", "0"):  # ("2": embedded also where the host packs the points, as in this run)
        os.environ["QSV_SHARE_CIRCUITS"] = share
        try:
            results.append(S.EVQEMinimumEigensolver(S.EVQEMinimumEigensolverConfiguration(**cfg)).compute_minimum_eigenvalue(
                OperatorCircuitEvaluator(op)))
        finally:
            del os.environ["QSV_SHARE_CIRCUITS"]
    assert results[0].eigenvalue == results[1].eigenvalue and results[0].best_individual == results[1].best_individual
    assert results[0].circuit_evaluations == results[1].circuit_evaluations
