"""JSSP domain-wall Hamiltonian builder: known energies from the reference's notebooks (re-derived by hand in
SURVEY.md section 6) and the exhaustive-state properties of the reference's encoder tests
(test/job_shop_scheduling/test_domain_wall_hamiltonian_encoder.py:28-124)."""

import numpy as np
import pytest

import jssp_instances as inst
from oracle import statevector_oracle as so
from queasars_amd import job_shop_scheduling as jss
from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder


def energies(encoder):
    op = encoder.get_problem_hamiltonian()
    assert op.is_diagonal() and op.num_qubits == encoder.n_qubits
    return so.diagonal_values(encoder.n_qubits, op.z_mask.tolist(), op.coeffs.real.tolist())


def energy_of(encoder, start_times_by_job):
    starts = {}
    for job, times in zip(encoder.jssp_instance.jobs, start_times_by_job):
        for operation, t in zip(job.operations, times):
            starts[operation] = t
    bitstring = encoder.bitstring_of(starts)
    result = encoder.translate_result_bitstring(bitstring)
    assert result.start_times == starts
    return energies(encoder)[int(bitstring, 2)], result


def test_notebook_energy_22_75():
    enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 12  # examples/evqe_jssp_optimization.ipynb:136
    value, result = energy_of(enc, [(0, 1, 2), (1, 3, 4)])
    assert result.is_valid and result.makespan == 5
    assert abs(value - 22.75) < 1e-9  # examples/evqe_jssp_optimization.ipynb:384-392
    all_e = energies(enc)
    valid_min = min(
        all_e[i] for i in range(1 << 12) if enc.translate_result_bitstring(format(i, "012b")).is_valid
    )
    assert abs(valid_min - 22.75) < 1e-9 and abs(all_e.min() - 22.75) < 1e-9


def test_runtime_notebook_energy_22_75():
    """The fourth energy the reference holds: examples/using_the_ibm_runtime.ipynb, cell 8's output ("Current best expectation
    value: 22.750000" from generation 1 on) on the 8-qubit "Simple Instance" of cell 2 with cell 6's encoder arguments; cell
    14 prints the makespan-4 schedule (j0: 1, 3; j1: 0, 1) as the most probable state of the result."""
    enc = JSSPDomainWallHamiltonianEncoder(inst.runtime_simple_instance(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 8  # cell 6: "needed qubits:  8"
    value, result = energy_of(enc, [(1, 3), (0, 1)])
    assert result.is_valid and result.makespan == 4
    assert abs(value - 22.75) < 1e-9
    all_e = energies(enc)
    assert abs(all_e.min() - 22.75) < 1e-9
    # the other schedule cell 14 prints (j0: 0, 3; j1: 0, 1) overlaps on m0: not valid, and dearer
    other, result = energy_of(enc, [(0, 3), (0, 1)])
    assert not result.is_valid and other > 22.75


def test_small_example_energies():
    enc = JSSPDomainWallHamiltonianEncoder(inst.small_2x2(), makespan_limit=3, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 4
    value, result = energy_of(enc, [(0, 1), (1, 2)])
    assert result.is_valid and result.makespan == 3 and abs(value - 63.5) < 1e-9
    assert abs(energies(enc).min() - 63.5) < 1e-9
    enc = JSSPDomainWallHamiltonianEncoder(inst.small_asymmetric(), makespan_limit=4, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 5
    value, result = energy_of(enc, [(1, 2), (0, 1, 2)])
    assert result.is_valid and result.makespan == 4 and abs(value - 61.6) < 1e-9
    assert abs(energies(enc).min() - 61.6) < 1e-9


def test_raises_for_too_small_timelimit():
    with pytest.raises(ValueError):
        JSSPDomainWallHamiltonianEncoder(inst.unit_test_instance(), makespan_limit=1).get_problem_hamiltonian()


def test_encoding_constraint_energy_level():
    penalty = 100
    enc = JSSPDomainWallHamiltonianEncoder(
        inst.unit_test_instance(), 4, encoding_penalty=penalty, overlap_constraint_penalty=0,
        precedence_constraint_penalty=0, max_opt_value=0,
    )
    e = energies(enc)
    for i in range(1 << enc.n_qubits):
        result = enc.translate_result_bitstring(format(i, f"0{enc.n_qubits}b"))
        if any(t is None for t in result.start_times.values()):
            assert e[i] >= penalty - 1e-9
        else:
            assert abs(e[i]) < 1e-9


def test_jssp_constraint_energy_level():
    penalty = 100
    enc = JSSPDomainWallHamiltonianEncoder(
        inst.unit_test_instance(), 4, encoding_penalty=0, overlap_constraint_penalty=penalty,
        precedence_constraint_penalty=penalty, max_opt_value=0,
    )
    e = energies(enc)
    for i in range(1 << enc.n_qubits):
        result = enc.translate_result_bitstring(format(i, f"0{enc.n_qubits}b"))
        encoded = all(t is not None for t in result.start_times.values())
        if encoded and not result.is_valid:
            assert e[i] >= penalty - 1e-9
        if result.is_valid:
            assert abs(e[i]) < 1e-9


def test_optimization_energy_level():
    enc = JSSPDomainWallHamiltonianEncoder(
        inst.unit_test_instance(), 4, encoding_penalty=0, overlap_constraint_penalty=0,
        precedence_constraint_penalty=0, max_opt_value=100, opt_all_operations_share=0,
    )
    e = energies(enc)
    by_makespan = {2: [], 3: [], 4: []}
    for i in range(1 << enc.n_qubits):
        result = enc.translate_result_bitstring(format(i, f"0{enc.n_qubits}b"))
        if result.is_valid:
            assert e[i] <= 100 + 1e-9
            by_makespan[result.makespan].append(e[i])
    assert max(by_makespan[2]) < min(by_makespan[3]) and max(by_makespan[3]) < min(by_makespan[4])


def test_three_by_three_instance_size():
    enc = JSSPDomainWallHamiltonianEncoder(inst.three_by_three(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
    assert enc.n_qubits == 18
    value, result = energy_of(enc, [(0, 1, 2), (0, 1, 2), (0, 1, 2)])
    assert result.is_valid and result.makespan == 3
    # makespan term: 3 jobs all ending at 3, norm 3 * 4^5; no early-start penalty
    assert abs(value - 100 * 0.81 * (3 * 4**3) / (3 * 4**5)) < 1e-9


def test_bitstring_round_trip_and_errors():
    enc = JSSPDomainWallHamiltonianEncoder(inst.small_2x2(), makespan_limit=3)
    with pytest.raises(ValueError):
        enc.translate_result_bitstring("101")
    with pytest.raises(ValueError):
        enc.translate_result_bitstring("10a1")
    assert enc.translate_result_bitstring("0101").start_times is not None


# ---- the reference's datatype tests (test/job_shop_scheduling/test_problem_instances.py) ----


def _two_ops():
    m1, m2 = jss.Machine("m1"), jss.Machine("m2")
    return m1, m2, jss.Operation("op1", "j1", m1, 1), jss.Operation("op2", "j1", m2, 2)


def test_reference_names_and_durations_are_checked():
    exc = jss.JobShopSchedulingProblemException
    m1, m2, op1, op2 = _two_ops()
    for make in (lambda: jss.Machine(""),
                 lambda: jss.Operation("", "test", m1, 5), lambda: jss.Operation("test", "", m1, 2),
                 lambda: jss.Operation("test", "test", m1, 0), lambda: jss.Operation("test", "test", m1, -5),
                 lambda: jss.Job("", (op1, op2)), lambda: jss.Job("test", ()),
                 lambda: jss.Job("j1", (op1, jss.Operation("op1", "j1", m2, 2))),       # duplicate identifiers
                 lambda: jss.Job("j1", (op1, jss.Operation("op2", "j2", m2, 2))),       # another job's operation
                 lambda: jss.Job("j1", (op1, jss.Operation("op2", "j1", m1, 2)))):      # a machine twice
        with pytest.raises(exc):
            make()


def test_reference_operation_identifiers():
    m1, m2 = jss.Machine("m1"), jss.Machine("m2")
    same = [jss.Operation("op1", "j1", m1, 3), jss.Operation("op1", "j1", m2, 3), jss.Operation("op1", "j1", m1, 4)]
    assert len({op.identifier for op in same}) == 1
    different = [jss.Operation("op1", "j1", m1, 3), jss.Operation("op2", "j1", m1, 3), jss.Operation("op1", "j2", m1, 3)]
    assert len({op.identifier for op in different}) == 3


def test_reference_instances_are_checked():
    exc = jss.JobShopSchedulingProblemException
    m1, m2, m3 = jss.Machine("m1"), jss.Machine("m2"), jss.Machine("m3")

    def single(job_name, machine):
        return jss.Job(job_name, (jss.Operation("op1", job_name, machine, 1),))

    job = jss.Job("j1", (jss.Operation("op1", "j1", m1, 1), jss.Operation("op2", "j1", m3, 2)))
    assert not job.is_consistent_with_machines((m1, m2)) and job.is_consistent_with_machines((m1, m2, m3))
    with pytest.raises(exc):
        jss.JobShopSchedulingProblemInstance("", (m1, m2), (single("j1", m1), single("j2", m2)))
    with pytest.raises(exc):
        jss.JobShopSchedulingProblemInstance("instance", (m1, jss.Machine("m1")), (single("j1", m1), single("j2", m1)))
    with pytest.raises(exc):
        jss.JobShopSchedulingProblemInstance("instance", (m1, m2), (single("j1", m1), single("j1", m2)))
    with pytest.raises(exc):
        jss.JobShopSchedulingProblemInstance("instance", (m1, m2), (single("j1", m1), single("j2", m3)))


def _small_instance():
    m1, m2 = jss.Machine("m1"), jss.Machine("m2")
    j1 = jss.Job("j1", (jss.Operation("op1", "j1", m1, 2), jss.Operation("op2", "j1", m2, 2)))
    j2 = jss.Job("j2", (jss.Operation("op3", "j2", m2, 2),))
    return jss.JobShopSchedulingProblemInstance("instance", (m1, m2), (j1, j2))


def _schedule(instance, starts):
    return {job: tuple(jss.ScheduledOperation(op, t) if t is not None else jss.UnscheduledOperation(op)
                       for op, t in zip(job.operations, times))
            for job, times in zip(instance.jobs, starts)}


def test_reference_scheduled_operations_and_results():
    op = jss.Operation("op", "job", jss.Machine("m"), 2)
    scheduled = jss.ScheduledOperation(operation=op, start_time=3)
    assert scheduled.is_scheduled and scheduled.end_time == 5
    assert not jss.UnscheduledOperation(operation=op).is_scheduled
    instance = _small_instance()
    good = jss.JobShopSchedulingResult(problem_instance=instance, schedule=_schedule(instance, [(0, 2), (0,)]))
    assert good.is_valid and good.makespan == 4 and good.valid_schedule == good.schedule
    for starts in ([(2, 1), (3,)],      # the operations of the first job overlap
                   [(0, 2), (1,)],      # two operations on the second machine at once
                   [(0, None), (0,)]):  # an operation without a start time
        bad = jss.JobShopSchedulingResult(instance, _schedule(instance, starts))
        assert not bad.is_valid and bad.makespan is None
        with pytest.raises(jss.JobShopSchedulingProblemException):
            bad.valid_schedule
    stranger = jss.Job("j", (jss.Operation("op", "j", jss.Machine("m1"), 2),))
    with pytest.raises(jss.JobShopSchedulingProblemException):
        jss.JobShopSchedulingResult(instance, {stranger: (jss.ScheduledOperation(stranger.operations[0], 0),)})


def test_reference_json_wire_format_round_trips():
    """test/job_shop_scheduling/test_serialization.py: instance, valid result, invalid result; and the key names of the
    reference's encoder (serialization.py:31-76), which make the files interchangeable."""
    import json

    instance = _small_instance()
    text = json.dumps(instance, cls=jss.JSSPJSONEncoder, indent=2)
    assert json.loads(text, cls=jss.JSSPJSONDecoder) == instance
    raw = json.loads(text)
    assert set(raw) == {"jssp_instance_name", "jssp_instance_machines", "jssp_instance_jobs"}
    first_job = raw["jssp_instance_jobs"]["tuple"][0]
    assert set(first_job) == {"job_name", "job_operations"}
    assert set(first_job["job_operations"]["tuple"][0]) == {"operation_name", "operation_job_name", "operation_machine",
                                                             "operation_processing_duration"}
    assert first_job["job_operations"]["tuple"][0]["operation_machine"] == {"machine_name": "m1"}
    for starts in ([(0, 2), (0,)], [(0, None), (1,)]):
        result = jss.JobShopSchedulingResult(instance, _schedule(instance, starts))
        text = json.dumps(result, cls=jss.JSSPJSONEncoder)
        back = json.loads(text, cls=jss.JSSPJSONDecoder)
        assert back.problem_instance == result.problem_instance and back.schedule == result.schedule and back == result
        raw = json.loads(text)
        assert set(raw) == {"jssp_result_problem_instance", "jssp_result_schedule"}
        entry = raw["jssp_result_schedule"]["dict"][0]["tuple"][1]["tuple"][1]
        assert set(entry) in ({"scheduled_operation", "scheduled_start_time"}, {"unscheduled_operation"})
    # an encoder's result travels through translate_result_bitstring as well
    enc = JSSPDomainWallHamiltonianEncoder(inst.notebook_2x3(), makespan_limit=6, **inst.NOTEBOOK_PENALTIES)
    decoded = enc.translate_result_bitstring("0" * enc.n_qubits)
    assert json.loads(json.dumps(decoded, cls=jss.JSSPJSONEncoder), cls=jss.JSSPJSONDecoder) == decoded


# ---- vectors the reference itself produced (tests/golden/make_jssp_golden.py; its JSSP datatype modules are plain Python) ----


def _reference_cases():
    import json
    from pathlib import Path

    return json.loads((Path(__file__).parent / "golden" / "jssp_reference.json").read_text())["cases"]


def _arguments(case):
    args = dict(case["arguments"])
    for key in ("relative_op_amount", "op_duration"):
        if isinstance(args[key], dict):
            args[key] = {k: v for k, v in args[key]["distribution"]}
    return args


def test_random_instances_are_the_references_for_the_same_seed():
    """random_job_shop_scheduling_instance consumes its generator as the reference's does: the same seed and arguments give
    the same machines, jobs, operations and durations; and this repository's encoder writes the reference's text byte for byte."""
    import json

    cases = _reference_cases()
    assert len(cases) >= 6
    for case in cases:
        instance = jss.random_job_shop_scheduling_instance(**_arguments(case))
        assert json.loads(json.dumps(case["wire"]), cls=jss.JSSPJSONDecoder) == instance
        assert json.dumps(instance, cls=jss.JSSPJSONEncoder, indent=2) == case["text_indent_2"]
        assert json.loads(case["text_indent_2"], cls=jss.JSSPJSONDecoder) == instance
        assert repr(instance) == case["repr"] and str(instance) == case["repr"]  # (what a notebook prints)
    with pytest.raises(ValueError):
        jss.random_job_shop_scheduling_instance("x", 2, 2, {0.5: 0.4, 1.0: 0.4}, 1, random_seed=0)


def test_schedule_validity_and_makespan_are_the_references():
    """48 schedules of the reference's random instances (sequential, per-job greedy, random, some operations left out): the
    reference's ``is_valid`` and ``makespan`` for each, and its wire form of the whole result."""
    import json

    n_valid = 0
    for case in _reference_cases():
        instance = json.loads(json.dumps(case["wire"]), cls=jss.JSSPJSONDecoder)
        for entry in case["schedules"]:
            result = jss.JobShopSchedulingResult(instance, _schedule(instance, entry["start_times"]))
            assert result.is_valid == entry["is_valid"] and result.makespan == entry["makespan"]
            n_valid += result.is_valid
            if "wire" in entry:
                assert json.loads(json.dumps(result, cls=jss.JSSPJSONEncoder)) == entry["wire"]
                assert json.loads(json.dumps(entry["wire"]), cls=jss.JSSPJSONDecoder) == result
                assert repr(result) == entry["repr"]
    assert n_valid >= 15
