"""Generates tests/golden/evqe_small.json from the two-formulation oracle.

The reference itself cannot produce vectors here (Qiskit is not installed), so these fixtures pin the oracle
against regressions and give the GPU tests fixed inputs/outputs; both oracle formulations must agree before a
case is written.  Run from the repository root:  python tests/golden/make_golden.py
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


if __name__ == "__main__":
    main()
