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
