"""Generates tests/golden/evqe_small.json from the two-formulation oracle.

The reference itself cannot produce vectors here (Qiskit is not installed), so these fixtures pin the oracle
against regressions and give the GPU tests fixed inputs/outputs; both oracle formulations must agree before a
case is written.  Run from the repository root:  python tests/golden/make_golden.py

    python tests/golden/make_golden.py --from-qiskit
        Where Qiskit (and Qiskit Aer) can be imported -- not in the build container --: recomputes every case's
        expectation value (and stored state) with the reference's own arithmetic -- ``EstimatorV2.run(pubs, precision=0)``
        as queasars/circuit_evaluation/circuit_evaluation.py:204-215 calls it, ``Statevector`` for the states -- compares
        with the stored oracle values and writes evqe_small_qiskit.json: the vectors that would PIN the oracle (SURVEY
        8(c): "to be confirmed on a Qiskit-equipped host").  Exits with status 3 when Qiskit is not importable.

    python tests/golden/make_golden.py --population
        Writes population_n6.json: a small EVQE population in the reference's wire format
        (queasars/minimum_eigensolvers/evqe/serialization.py:27-65) with the oracle's expectation values beside it.
"""

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import helpers  # noqa: E402
from oracle import statevector_oracle as so  # noqa: E402


def main():
    cases = []
    for n in (2, 3, 4, 5, 6, 8, 10):
        for layers in (1, 2, 4):
            if n == 10 and layers == 4:
                continue
            seed = 100 * n + layers
            _, circuits, params = helpers.population_circuits(n, layers, 1, seed=seed)
            op = helpers.random_pauli_operator(n, min(20, 4**n - 1), seed=seed + 1)
            ops = circuits[0].bound_ops(params[0])
            state = so.simulate(n, ops)
            dense = so.simulate_dense(n, ops)
            assert np.abs(state - dense).max() < 1e-13
            e = so.pauli_expectation(state, op.x_mask.tolist(), op.z_mask.tolist(), op.coeffs.tolist())
            e2 = so.pauli_expectation_dense(dense, op.labels, op.coeffs.tolist())
            assert abs(e - e2) < 1e-13
            case = {
                "n_qubits": n,
                "n_layers": layers,
                "seed": seed,
                "ops": [list(o) for o in ops],
                "labels": op.labels,
                "coeffs": op.coeffs.real.tolist(),
                "expectation": e.real,
            }
            if n <= 6:
                case["state_re"] = state.real.tolist()
                case["state_im"] = state.imag.tolist()
            cases.append(case)
    out = Path(__file__).parent / "evqe_small.json"
    out.write_text(json.dumps({"generator": "tests/golden/make_golden.py", "cases": cases}, indent=0))
    print(f"wrote {len(cases)} cases to {out}")


def from_qiskit():
    try:
        from qiskit import QuantumCircuit
        from qiskit.circuit.library import CU3Gate
        from qiskit.quantum_info import SparsePauliOp, Statevector
    except Exception as exc:  # not installable offline
        print(f"Qiskit is not importable here ({type(exc).__name__}: {exc}); nothing written")
        return 3
    try:
        from qiskit_aer.primitives import EstimatorV2
    except Exception:
        from qiskit.primitives import StatevectorEstimator as EstimatorV2  # the reference accepts any EstimatorV2
    data = json.loads((Path(__file__).parent / "evqe_small.json").read_text())
    estimator = EstimatorV2()
    worst_e = worst_s = 0.0
    out_cases = []
    for case in data["cases"]:
        qc = QuantumCircuit(case["n_qubits"])
        for kind, target, control, theta, phi, lam in case["ops"]:
            if kind == 0:
                qc.id(target)
            elif kind == 1:
                qc.u(theta, phi, lam, target)  # quantum_gate.py:96-102
            else:
                qc.append(CU3Gate(theta, phi, lam), [control, target])  # quantum_gate.py:157-165
        op = SparsePauliOp(case["labels"], case["coeffs"])
        value = float(np.real(estimator.run([(qc, op)], precision=0).result()[0].data.evs))
        worst_e = max(worst_e, abs(value - case["expectation"]))
        new = dict(case, expectation=value, source="qiskit")
        if "state_re" in case:
            state = np.asarray(Statevector(qc).data)
            worst_s = max(worst_s, float(np.abs(state - (np.asarray(case["state_re"]) + 1j * np.asarray(case["state_im"]))).max()))
            new["state_re"], new["state_im"] = state.real.tolist(), state.imag.tolist()
        out_cases.append(new)
    print(f"max |dE| oracle vs Qiskit = {worst_e:.3e}, max |d amplitude| = {worst_s:.3e} over {len(out_cases)} cases")
    out = Path(__file__).parent / "evqe_small_qiskit.json"
    out.write_text(json.dumps({"generator": "tests/golden/make_golden.py --from-qiskit", "cases": out_cases}, indent=0))
    print(f"wrote {out}")
    return 0 if worst_e < 1e-10 and worst_s < 1e-12 else 1


def population_fixture():
    from queasars_amd.evqe import EVQEPopulation
    from queasars_amd.evqe.serialization import population_to_dict

    n, layers, count = 6, 3, 5
    population = EVQEPopulation.random_population(n, layers, count, True, 606)
    op = helpers.random_ising_operator(n, seed=66)
    values = []
    for ind in population.individuals:
        c = ind.get_parameterized_quantum_circuit()
        values.append(helpers.oracle_expectation(c, list(ind.parameter_values), op))
    out = Path(__file__).parent / "population_n6.json"
    out.write_text(json.dumps({
        "generator": "tests/golden/make_golden.py --population",
        "population": population_to_dict(population),
        "operator": {"labels": op.labels, "coeffs": op.coeffs.real.tolist()},
        "expectations": values,
    }, indent=0))
    print(f"wrote {out}")


if __name__ == "__main__":
    if "--from-qiskit" in sys.argv[1:]:
        sys.exit(from_qiskit())
    if "--population" in sys.argv[1:]:
        population_fixture()
    else:
        main()
