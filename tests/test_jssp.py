"""JSSP domain-wall Hamiltonian builder: known energies from the reference's notebooks (re-derived by hand in
SURVEY.md section 6) and the exhaustive-state properties of the reference's encoder tests
(test/job_shop_scheduling/test_domain_wall_hamiltonian_encoder.py:28-124)."""

import numpy as np
import pytest

import jssp_instances as inst
from oracle import statevector_oracle as so
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
