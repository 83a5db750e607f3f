"""CPU oracle for the circuit-evaluation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``queasars_amd``) never imports this package and fails loudly when its HIP
library is missing.

PARITY UNPINNED.  The arithmetic of the reference's hot path lives in
third-party Qiskit / Qiskit Aer (qiskit 2.4.2, qiskit-aer 0.17.2,
qiskit-algorithms 0.4.0 per the reference's poetry.lock), which are neither
vendored in the reference nor installed here, and the reference's own tests
hold no numeric golden vector for this path.  The oracle therefore restates
the *published* gate / Pauli / binding semantics and is pinned only by
analytic known answers (see ``tests/test_oracle.py``) and by agreement
between its independent formulations.
"""
