"""Host-side pieces of the evaluator API that need no GPU: IR, operator format, CVaR post-processing,
bitstring evaluator (mirrors what the reference's circuit_evaluation package does outside the primitive)."""

import pickle

import numpy as np
import pytest

from oracle import statevector_oracle as so
from queasars_amd.circuit_evaluation import (
    BitstringEvaluator,
    BitstringEvaluatorException,
    get_expectation_with_bitstring_evaluator,
    get_expectation_with_operator,
)
from queasars_amd.circuit_evaluation.expectation_calculation import basis_state_values
from queasars_amd.ir import NO_CONTROL, OP_CU3, OP_ID, OP_U, CircuitIR, ParamRef, PauliOperator


class TestCircuitIR:
    def test_builder_and_packing(self):
        c = CircuitIR(3).id(0).u(0.1, ParamRef(1), 0.3, 2).cu3(ParamRef(0), 0.5, ParamRef(2), 0, 1)
        ops = c.packed()
        assert len(c) == 3 and c.num_parameters == 3 and c.count_ops() == {"id": 1, "u": 1, "cu3": 1}
        assert list(ops["kind"]) == [OP_ID, OP_U, OP_CU3]
        assert list(ops["target"]) == [0, 2, 1] and list(ops["control"]) == [NO_CONTROL, NO_CONTROL, 0]
        assert list(ops["p_phi"]) == [-1, 1, -1] and ops["theta"][1] == 0.1
        assert c.bound_ops([9.0, 8.0, 7.0]) == [(0, 0, -1, 0.0, 0.0, 0.0), (1, 2, -1, 0.1, 8.0, 0.3), (2, 1, 0, 9.0, 0.5, 7.0)]

    def test_argument_errors(self):
        with pytest.raises(ValueError):
            CircuitIR(0)
        with pytest.raises(ValueError):
            CircuitIR(2).u(0, 0, 0, 2)
        with pytest.raises(ValueError):
            CircuitIR(2).cu3(0, 0, 0, 1, 1)
        with pytest.raises(ValueError):
            CircuitIR(2).u(ParamRef(0), 0, 0, 0).bound_ops([])
        with pytest.raises(ValueError):
            CircuitIR(2).compose(CircuitIR(3))

    def test_compose_keeps_order(self):
        a, b = CircuitIR(2).u(0.1, 0.2, 0.3, 0), CircuitIR(2).cu3(0.4, 0.5, 0.6, 0, 1)
        assert [o[0] for o in a.compose(b).bound_ops([])] == [OP_U, OP_CU3]


class TestPauliOperator:
    def test_masks_follow_qiskit_label_order(self):
        op = PauliOperator(["IZ", "XI", "YZ"], [1.0, 2.0, 3.0])
        assert op.x_mask.tolist() == [0, 2, 2] and op.z_mask.tolist() == [1, 0, 3]
        assert op.num_qubits == 2 and len(op) == 3 and not op.is_diagonal()
        assert PauliOperator(["ZI", "II"]).is_diagonal()

    def test_sparse_list_and_pickle(self):
        op = PauliOperator.from_sparse_list([("ZZ", [0, 3], 0.5), ("X", [1], -1.0)], 4)
        assert op.labels == ["ZIIZ", "IIXI"]
        clone = pickle.loads(pickle.dumps(op))
        assert clone.labels == op.labels and np.array_equal(clone.coeffs, op.coeffs)

    def test_bad_labels(self):
        with pytest.raises(ValueError):
            PauliOperator([])
        with pytest.raises(ValueError):
            PauliOperator(["ZZ", "Z"])
        with pytest.raises(ValueError):
            PauliOperator(["ZQ"])
        with pytest.raises(ValueError):
            PauliOperator(["ZZ"], [1.0, 2.0])


class TestSamplerPostProcessing:
    def test_values_match_the_oracle(self):
        rng = np.random.default_rng(0)
        op = PauliOperator.from_sparse_list([("ZZ", [0, 2], 0.7), ("Z", [1], -1.3), ("ZZ", [1, 4], 0.2)], 5)
        states = rng.integers(0, 32, size=20)
        want = [so.evaluate_sparsepauli(int(s), op.z_mask.tolist(), op.coeffs.tolist()).real for s in states]
        assert np.allclose(basis_state_values(states, op), want, atol=1e-15)

    @pytest.mark.parametrize("alpha", [1.0, 0.5, 0.3, 0.05])
    def test_cvar_matches_the_oracle(self, alpha):
        rng = np.random.default_rng(3)
        op = PauliOperator.from_sparse_list([("ZZ", [0, 1], 1.0), ("Z", [2], 0.5), ("Z", [0], -0.25)], 3)
        counts = rng.integers(1, 50, size=8)
        dist = {s: c / counts.sum() for s, c in enumerate(counts)}
        got = get_expectation_with_operator(dist, op, alpha)
        want = so.expectation_from_distribution(dist, op.z_mask.tolist(), op.coeffs.tolist(), alpha)
        assert abs(got - want) < 1e-14

    def test_alpha_range_and_diagonal_requirement(self):
        op = PauliOperator(["ZI"])
        for bad in (0.0, -0.1, 1.5):
            with pytest.raises(ValueError):
                get_expectation_with_operator({0: 1.0}, op, bad)
        with pytest.raises(ValueError):
            get_expectation_with_operator({0: 1.0}, PauliOperator(["XI"]), 1.0)

    def test_bitstring_evaluator(self):
        ev = BitstringEvaluator(3, lambda b: float(int(b, 2)))
        assert ev.evaluate_bitstring("101") == 5.0 and ev.input_length == 3
        with pytest.raises(BitstringEvaluatorException):
            ev.evaluate_bitstring("10")
        with pytest.raises(BitstringEvaluatorException):
            ev.evaluate_bitstring("1a1")
        dist = {0b001: 0.5, 0b110: 0.5}
        assert get_expectation_with_bitstring_evaluator(dist, ev, 1.0) == 0.5 * 1 + 0.5 * 6
        assert get_expectation_with_bitstring_evaluator(dist, ev, 0.5) == 1.0


class TestCoalescing:
    """queasars_amd.circuit_evaluation.coalescing: concurrent one-circuit calls are answered from merged batches."""

    class _Fake:
        n_qubits = 3

        def __init__(self, fail_on=None):
            self.calls = []
            self.fail_on = fail_on

        def evaluate_circuits(self, circuits, parameter_values):
            import time as _t

            self.calls.append(len(circuits))
            _t.sleep(0.002)  # callers pile up meanwhile
            if self.fail_on is not None and self.fail_on in circuits:
                raise RuntimeError("boom")
            return [float(c) + sum(p) for c, p in zip(circuits, parameter_values)]

    def test_merges_concurrent_calls_and_keeps_every_callers_order(self):
        from concurrent.futures import ThreadPoolExecutor

        from queasars_amd.circuit_evaluation import CoalescingCircuitEvaluator

        fake = self._Fake()
        ev = CoalescingCircuitEvaluator(fake, window_s=1e-3)
        assert ev.n_qubits == 3
        jobs = [([10 * i, 10 * i + 1], [[0.5], [0.25, 0.25]]) for i in range(24)]
        with ThreadPoolExecutor(max_workers=24) as pool:
            got = list(pool.map(lambda j: ev.evaluate_circuits(*j), jobs))
        assert got == [[10 * i + 0.5, 10 * i + 1 + 0.5] for i in range(24)]
        assert sum(fake.calls) == 48 and len(fake.calls) < 24 and ev.n_batches == len(fake.calls)
        assert ev.evaluate_circuits([], []) == []
        assert ev.evaluate_circuits([7], [[1.0]]) == [8.0]  # a lone caller is its own leader

    def test_failure_reaches_every_caller_of_the_batch_and_the_evaluator_recovers(self):
        from concurrent.futures import ThreadPoolExecutor

        import pytest as _pytest

        from queasars_amd.circuit_evaluation import CoalescingCircuitEvaluator

        fake = self._Fake(fail_on=13)
        ev = CoalescingCircuitEvaluator(fake, window_s=2e-3)
        with ThreadPoolExecutor(max_workers=4) as pool:
            futures = [pool.submit(ev.evaluate_circuits, [c], [[0.0]]) for c in (11, 12, 13, 14)]
            errors = [f.exception() for f in futures]
        assert any(isinstance(e, RuntimeError) for e in errors)
        fake.fail_on = None
        assert ev.evaluate_circuits([5], [[0.5]]) == [5.5]
        with _pytest.raises(ValueError):
            ev.evaluate_circuits([1], [])


class TestQiskitAdapter:
    """queasars_amd.qiskit_adapter is duck-typed: stand-ins with the attributes of QuantumCircuit / SparsePauliOp that it
    documents are enough to exercise it without Qiskit."""

    class _Param:
        def __init__(self, name):
            self.name = name
            self.parameters = {self}

        def __str__(self):
            return self.name

        def __hash__(self):
            return hash(self.name)

        def __eq__(self, other):
            return isinstance(other, type(self)) and other.name == self.name

    class _Circuit:
        def __init__(self, n, data, parameters):
            from types import SimpleNamespace

            self.num_qubits = n
            self.parameters = parameters
            self.data = [SimpleNamespace(operation=SimpleNamespace(name=name, params=params), qubits=qubits) for name, params, qubits in data]

        def find_bit(self, q):
            from types import SimpleNamespace

            return SimpleNamespace(index=q)

    def test_lowering_follows_name_sorted_parameters_and_control_target_order(self):
        from queasars_amd import qiskit_adapter
        from queasars_amd.ir import CircuitIR, ParamRef

        P = self._Param
        # Qiskit keeps circuit.parameters sorted by name: layer0_q1_* before layer0_q2_*
        a, b, c = P("layer0_q1_lambda"), P("layer0_q1_phi"), P("layer0_q1_theta")
        d = P("layer0_q2_theta")
        qc = self._Circuit(
            3,
            [("id", [], [0]), ("u", [c, b, a], [1]), ("cu3", [d, 0.25, 0.5], [1, 2]), ("barrier", [], [0, 1, 2]), ("measure", [], [0])],
            [a, b, c, d],
        )
        got = qiskit_adapter.circuit_from_qiskit(qc)
        want = CircuitIR(3).id(0).u(ParamRef(2), ParamRef(1), ParamRef(0), 1).cu3(ParamRef(3), 0.25, 0.5, 1, 2)
        assert got.num_parameters == 4 and got.packed().tobytes() == want.packed().tobytes()

    def test_rejects_what_it_cannot_represent(self):
        import pytest as _pytest

        from queasars_amd import qiskit_adapter

        with _pytest.raises(ValueError, match="unsupported instruction"):
            qiskit_adapter.circuit_from_qiskit(self._Circuit(2, [("cx", [], [0, 1])], []))

        class _Expr:
            def __init__(self, p):
                self.parameters = {p}

            def __str__(self):
                return "2*x"

        with _pytest.raises(ValueError, match="expression"):
            qiskit_adapter.circuit_from_qiskit(self._Circuit(1, [("u", [_Expr(self._Param("x")), 0.0, 0.0], [0])], [self._Param("x")]))

    def test_operator_labels_keep_qiskits_order(self):
        from types import SimpleNamespace

        from queasars_amd import qiskit_adapter

        op = SimpleNamespace(paulis=SimpleNamespace(to_labels=lambda: ["IZ", "XI"]), coeffs=[0.5, -1.0])
        got = qiskit_adapter.operator_from_qiskit(op)
        assert got.labels == ["IZ", "XI"] and got.z_mask.tolist() == [1, 0] and got.x_mask.tolist() == [0, 2]


class TestProcessLocalRegistrations:
    """A circuit's device registrations are meaningful only in the process (and for the device) that made them
    (advisor finding, round 1): copies and pickles start without them."""

    def test_pickle_and_copies_drop_the_registration(self):
        import copy

        c = CircuitIR(4).u(0.1, ParamRef(0), 0.3, 1).cu3(0.2, 0.3, ParamRef(1), 0, 2)
        c.packed()
        c._registered[("token", 1)] = 5
        for clone in (pickle.loads(pickle.dumps(c)), copy.deepcopy(c), copy.copy(c)):
            assert clone._registered == {} and clone.bound_ops([1.0, 2.0]) == c.bound_ops([1.0, 2.0])
            assert clone.num_parameters == 2 and np.array_equal(clone.packed(), c.packed())
        assert c._registered == {("token", 1): 5}
        clone = copy.deepcopy(c)
        clone.u(0.5, 0.5, 0.5, 3)  # editing a copy leaves the original (and its registration) alone
        assert len(c) == 2 and c._registered

    def test_device_keys_differ_between_processes(self):
        """The key a device files its registrations under carries a per-process random token: a circuit registered in
        one process and shipped to another (where device serials restart at 1) cannot be taken for registered there."""
        import subprocess
        import sys

        code = "from queasars_amd.circuit_evaluation import circuit_evaluation as ce; print(ce._PROCESS_TOKEN)"
        tokens = {subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip() for _ in range(2)}
        from queasars_amd.circuit_evaluation import circuit_evaluation as ce

        assert len(tokens) == 2 and ce._PROCESS_TOKEN not in tokens

    def test_finalizers_only_queue_ids(self):
        """weakref finalizers may run inside any allocation, also while this thread holds the handle between
        qsv_eval_begin and qsv_eval_end: they must not call into the library."""
        import gc

        from queasars_amd.circuit_evaluation.circuit_evaluation import StatevectorDevice

        class FakeLib:
            destroyed = []

            def qsv_circuit_destroy(self, handle, cid):
                self.destroyed.append(cid)
                return 0

        dev = object.__new__(StatevectorDevice)  # no GPU here: only the bookkeeping is exercised
        from queasars_amd.circuit_evaluation.circuit_evaluation import _make_gone

        dev._dead, dev._dead_states, dev._watched, dev._handle, dev._lib = [], [], {}, 1, FakeLib()
        dev._gone = _make_gone(dev._dead, dev._watched)
        c = CircuitIR(2).u(0.1, 0.2, 0.3, 0)
        dev._watch(c, 17)
        assert [cid for _, cid in dev._watched.values()] == [17]
        del c
        gc.collect()
        assert dev._dead == [17] and dev._watched == {} and FakeLib.destroyed == []
        dev._reap()
        assert dev._dead == [] and FakeLib.destroyed == [17]
        dev._handle = None  # keep __del__ from calling qsv_destroy on the fake


class TestUnpinnedCaches:
    def test_composed_circuits_follow_edits_and_do_not_pin(self):
        import gc

        from queasars_amd.circuit_evaluation.circuit_evaluation import _ComposedCircuits

        init = CircuitIR(3).u(0.3, 0.0, 0.0, 0)
        cache = _ComposedCircuits(init, limit=8)
        c = CircuitIR(3).cu3(0.1, 0.2, 0.3, 0, 1)
        first = cache.get(c)
        assert cache.get(c) is first and len(first) == 2
        c.u(0.4, 0.0, 0.0, 2)  # edited in place: the composed circuit must be rebuilt
        second = cache.get(c)
        assert second is not first and len(second) == 3
        del c, first, second
        gc.collect()
        assert len(cache) == 0
        keep = [CircuitIR(3).id(0) for _ in range(20)]
        for k in keep:
            cache.get(k)
        assert len(cache) <= 8
        assert _ComposedCircuits(None).get(keep[0]) is keep[0]

    def test_identity_cache_of_the_primitive_front_ends(self):
        import gc

        from queasars_amd.primitives import _IdentityCache

        class Foreign:
            pass

        calls = []
        cache = _IdentityCache(lambda obj: calls.append(id(obj)) or len(calls), limit=4)
        a = Foreign()
        assert cache.get(a) == cache.get(a) == 1 and len(calls) == 1
        del a
        gc.collect()
        assert len(cache) == 0
        tuples = [(i,) for i in range(10)]  # tuples cannot be weakly referenced: bounded strong table
        for t in tuples:
            cache.get(t)
        assert len(cache) <= 4

    def test_vectorised_cvar_equals_the_reference_loop(self):
        """The product's vectorised CVaR accumulation against the oracle's literal restatement of the reference's
        loop (expectation_calculation.py:14-32), including distributions whose mass falls short of alpha."""
        from queasars_amd.circuit_evaluation.expectation_calculation import _get_expectation

        rng = np.random.default_rng(11)
        for trial in range(200):
            m = int(rng.integers(1, 12))
            p = rng.random(m)
            p = p / p.sum() * (1.0 if trial % 3 else 0.9)
            values = rng.normal(size=m).round(1 if trial % 2 else 6)  # ties now and then
            alpha = float(rng.choice([1.0, 0.95, 0.5, 0.2, 0.01]))
            items = [(i, float(pi), float(vi)) for i, (pi, vi) in enumerate(zip(p, values))]
            assert abs(_get_expectation(items, alpha) - so.cvar_expectation(items, alpha)) < 1e-12

    def test_composed_circuit_cache_pickles_empty(self):
        from queasars_amd.circuit_evaluation.circuit_evaluation import _ComposedCircuits

        cache = _ComposedCircuits(CircuitIR(2).u(0.1, 0.2, 0.3, 0), limit=16)
        c = CircuitIR(2).id(1)
        cache.get(c)
        clone = pickle.loads(pickle.dumps(cache))
        assert len(clone) == 0 and len(clone.get(c)) == 2


def test_cvar_of_a_sample_matrix_equals_the_row_by_row_form():
    from queasars_amd.circuit_evaluation.circuit_evaluation import _cvar_of_sample_matrix, _cvar_of_samples

    rng = np.random.default_rng(5)
    values = rng.normal(size=(7, 512)).round(2)
    for alpha in (1.0, 0.5, 0.3, 0.123, 1 / 512, 0.999):
        want = [_cvar_of_samples(row, alpha) for row in values]
        got = _cvar_of_sample_matrix(values, alpha)
        assert np.allclose(got, want, rtol=0, atol=1e-12)
    assert _cvar_of_sample_matrix(np.zeros((0, 512)), 0.5) == []


class TestConfiguredPrimitives:
    """queasars_amd.circuit_evaluation.configured_primitives (reference: configured_primitives.py:9-22)."""

    def test_the_two_pairs_hold_their_option_and_refuse_nonsense(self):
        from queasars_amd.circuit_evaluation import ConfiguredEstimatorV2, ConfiguredSamplerV2

        sampler, estimator = object(), object()
        s = ConfiguredSamplerV2(sampler=sampler, shots=128)
        e = ConfiguredEstimatorV2(estimator=estimator, precision=0.0)
        assert s.sampler is sampler and s.shots == 128 and e.estimator is estimator and e.precision == 0.0
        assert s == ConfiguredSamplerV2(sampler, 128)  # plain dataclasses, as in the reference
        with pytest.raises(ValueError):
            ConfiguredSamplerV2(sampler, 0)
        with pytest.raises(ValueError):
            ConfiguredEstimatorV2(estimator, -1e-3)

    def test_evaluator_for_checks_its_arguments_before_touching_a_device(self):
        from queasars_amd.circuit_evaluation import ConfiguredEstimatorV2, ConfiguredSamplerV2, evaluator_for

        op = PauliOperator(["ZI"])
        ev = BitstringEvaluator(2, lambda b: 0.0)
        with pytest.raises(ValueError):
            evaluator_for(ConfiguredSamplerV2(object(), 8))
        with pytest.raises(ValueError):
            evaluator_for(ConfiguredSamplerV2(object(), 8), operator=op, bitstring_evaluator=ev)
        with pytest.raises(ValueError):
            evaluator_for(ConfiguredEstimatorV2(object(), 0.0), bitstring_evaluator=ev)
        with pytest.raises(TypeError):
            evaluator_for(object(), operator=op)


class TestSharedDeviceScratch:
    """Advisor finding (round 2): the scratch buffer of a batch was sized from a field another thread could overwrite
    between the metadata call and the allocation; the CPython-API packer then wrote past it."""

    @staticmethod
    def _bare_device():
        import threading

        from queasars_amd.circuit_evaluation.circuit_evaluation import StatevectorDevice, _make_gone

        dev = object.__new__(StatevectorDevice)  # no GPU here: only the bookkeeping is exercised
        dev._dead, dev._watched, dev._handle, dev._lib = [], {}, None, None
        dev._gone = _make_gone(dev._dead, dev._watched)
        dev._serial, dev._n_qubits, dev._last_batch, dev._reg_lock = ("test", 1), 3, None, threading.Lock()
        return dev

    def test_two_threads_with_batches_of_different_sizes_get_their_own_totals(self):
        import threading

        dev = self._bare_device()
        small = [CircuitIR(3).u(ParamRef(0), 0.1, 0.2, 0) for _ in range(2)]
        large = [CircuitIR(3).u(ParamRef(0), ParamRef(1), ParamRef(2), 1).cu3(ParamRef(3), ParamRef(4), ParamRef(5), 0, 2) for _ in range(9)]
        for i, c in enumerate(small + large):
            c.packed()
            c._registered[dev._serial] = i + 1  # (as if registered: no library call is made)
        wrong = []

        def worker(batch, want):
            for _ in range(2000):
                ids, need, total = dev._batch_metadata(batch)
                if total != int(need.sum()) or total != want or len(ids) != len(batch):
                    wrong.append((total, int(need.sum()), want))

        threads = [threading.Thread(target=worker, args=(small, 2)), threading.Thread(target=worker, args=(large, 54))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert wrong == []

    def test_the_packer_refuses_to_write_past_its_buffer(self):
        import ctypes as C

        from queasars_amd.circuit_evaluation.circuit_evaluation import _load_pyhelp

        helper = _load_pyhelp()
        if helper is None:
            pytest.skip("csrc/pyhelp.c was not built (no Python headers)")
        vectors = [[0.5] * 6, [1.5] * 6, [2.5] * 6]
        take = np.asarray([6, 4, 6], dtype=np.int64)
        out = np.full(32, -1.0)
        n = helper.qsv_pack_exact(vectors, 0, 3, take.ctypes.data, out.ctypes.data, 16)
        assert n == 16 and out[:16].tolist() == [0.5] * 6 + [1.5] * 4 + [2.5] * 6 and (out[16:] == -1.0).all()
        out[:] = -1.0
        with pytest.raises(ValueError, match="scratch"):
            helper.qsv_pack_exact(vectors, 0, 3, take.ctypes.data, out.ctypes.data, 15)
        assert (out[15:] == -1.0).all(), "nothing may be written beyond the capacity"
        with pytest.raises(ValueError, match="needs 6 parameter values, got 2"):
            helper.qsv_pack_exact([[1.0, 2.0]], 0, 1, take.ctypes.data, out.ctypes.data, 32)


def test_packer_takes_numpy_rows_as_they_stand():
    """Parameter vectors that are contiguous float64 buffers (rows of the optimiser's matrix) are copied, lists are
    unpacked value by value; a short row is refused like a short list; None among the vectors is found by identity."""
    from queasars_amd.circuit_evaluation.circuit_evaluation import _has_none, _load_pyhelp

    helper = _load_pyhelp()
    if helper is None:
        pytest.skip("helper not built")
    matrix = np.arange(40, dtype=np.float64).reshape(4, 10)
    vectors = [matrix[0, :7], list(matrix[1]), matrix[2], np.float32(matrix[3]), tuple(matrix[0])]
    take = np.array([7, 10, 4, 10, 2], dtype=np.int64)
    out = np.full(40, -1.0)
    n = helper.qsv_pack_exact(vectors, 0, 5, take.ctypes.data, out.ctypes.data, 40)
    want = np.concatenate([matrix[0, :7], matrix[1], matrix[2, :4], matrix[3], matrix[0, :2]])
    assert n == 33 and np.array_equal(out[:33], want) and (out[33:] == -1.0).all()
    with pytest.raises(ValueError):
        helper.qsv_pack_exact([matrix[0, :3]], 0, 1, np.array([5], dtype=np.int64).ctypes.data, out.ctypes.data, 40)
    strided = np.arange(20, dtype=np.float64)[::2]  # (not contiguous: goes the slow way, still right)
    assert helper.qsv_pack_exact([strided], 0, 1, np.array([10], dtype=np.int64).ctypes.data, out.ctypes.data, 40) == 10
    assert np.array_equal(out[:10], strided)
    assert not _has_none([matrix[0], matrix[1]]) and _has_none([matrix[0], None]) and not _has_none([])


def test_bitstring_evaluator_answers_as_the_references_does():
    """Values and refusals of the reference's own BitstringEvaluator (tests/golden/make_host_golden.py ran
    queasars/circuit_evaluation/bitstring_evaluation.py), among them a wrong length, other characters and a full-width digit."""
    import json
    from pathlib import Path

    from queasars_amd.circuit_evaluation.bitstring_evaluation import BitstringEvaluator, BitstringEvaluatorException

    data = json.loads((Path(__file__).parent / "golden" / "spsa_termination_reference.json").read_text())["bitstring_evaluator"]
    evaluator = BitstringEvaluator(data["input_length"], lambda bits: float(int(bits, 2)) / 4 - bits.count("1"))
    assert evaluator.input_length == data["input_length"]
    for case in data["cases"]:
        if "raises" in case:
            with pytest.raises(BitstringEvaluatorException):
                evaluator.evaluate_bitstring(case["bitstring"])
        else:
            assert evaluator.evaluate_bitstring(case["bitstring"]) == case["value"]
