"""Coalescing of concurrent evaluator calls.

The reference's EVQE calls ``evaluate_circuits`` from up to ``population_size`` threads at once, one or two
circuits per call (evqe.py:232-236, selection.py:75-85, mutation.py:63-75), and for simulators that are not
thread-safe puts a batching runner with a 0.1 s collection window in front of the primitive
(circuit_evaluation/mutex_primitives.py:67-199).  A GPU evaluation of one circuit is launch bound (tens of
microseconds) while a population of 64 costs half a millisecond in one call, so the same idea pays here with a
window three orders of magnitude shorter: callers that arrive while a batch is being collected or evaluated are merged
into the next one.

Scheme: the first caller to find no collector becomes the leader, waits ``window_s`` (other callers append their
circuits meanwhile), takes everything queued, evaluates it in ONE call of the wrapped evaluator and hands every caller
its slice.  Callers arriving during that evaluation queue up and elect a new leader when it finishes.
"""

from __future__ import annotations

import threading
import time
from typing import Optional, Sequence

from queasars_amd.circuit_evaluation.circuit_evaluation import BaseCircuitEvaluator


class _Ticket:
    __slots__ = ("circuits", "parameter_values", "done", "result", "error")

    def __init__(self, circuits, parameter_values):
        self.circuits = list(circuits)
        self.parameter_values = list(parameter_values)
        self.done = threading.Event()
        self.result: Optional[list[float]] = None
        self.error: Optional[BaseException] = None


class CoalescingCircuitEvaluator(BaseCircuitEvaluator):
    """Wraps any :class:`BaseCircuitEvaluator`; concurrent calls are answered from merged batches, in input order.

    One-circuit calls on an exact :class:`OperatorCircuitEvaluator` (no initial state, no noise) take the native path:
    ``qsv_eval_coalesced`` merges the callers inside the library while their threads wait in C with the GIL released
    (the Python scheme below costs several thread hand-overs under the GIL per call).  The wrapped evaluator's device
    must then not be shared with evaluators of other operators."""

    def __init__(self, evaluator: BaseCircuitEvaluator, window_s: float = 2e-4, max_batch: int = 4096, native: bool = True):
        if window_s < 0 or max_batch < 1:
            raise ValueError("window_s must be >= 0 and max_batch >= 1")
        self._evaluator = evaluator
        self._native = bool(native) and self._native_capable(evaluator)
        self._window_s = float(window_s)
        self._max_batch = int(max_batch)
        self._lock = threading.Lock()
        self._queue: list[_Ticket] = []
        self._leader_active = False
        self.n_batches = 0  # merged calls issued so far (for tests and tuning)

    @property
    def n_qubits(self) -> int:
        return self._evaluator.n_qubits

    @staticmethod
    def _native_capable(evaluator) -> bool:
        from queasars_amd.circuit_evaluation.circuit_evaluation import OperatorCircuitEvaluator

        return (type(evaluator) is OperatorCircuitEvaluator and evaluator._initial_state_circuit is None
                and evaluator._precision == 0)

    def evaluate_circuits(self, circuits: Sequence, parameter_values: Sequence[Sequence[float]]) -> list[float]:
        if len(circuits) != len(parameter_values):
            raise ValueError("circuits and parameter_values must have the same length")
        if not circuits:
            return []
        if self._native and len(circuits) == 1 and circuits[0] is not None and parameter_values[0] is not None:
            device = self._evaluator._device
            if device._operator is self._evaluator._operator:
                return [device.expectation_value_coalesced(circuits[0], parameter_values[0], self._window_s * 1e6)]
        ticket = _Ticket(circuits, parameter_values)
        with self._lock:
            self._queue.append(ticket)
            lead = not self._leader_active
            if lead:
                self._leader_active = True
        if lead:
            self._lead()
        ticket.done.wait()
        if ticket.error is not None:
            raise ticket.error
        return ticket.result

    def _lead(self) -> None:
        """Collect for one window, evaluate, repeat while callers keep queueing; then step down."""
        while True:
            if self._window_s:
                time.sleep(self._window_s)
            with self._lock:
                batch, total = [], 0
                while self._queue and (not batch or total + len(self._queue[0].circuits) <= self._max_batch):
                    t = self._queue.pop(0)
                    batch.append(t)
                    total += len(t.circuits)
                if not batch:
                    self._leader_active = False
                    return
            self._run(batch)

    def _run(self, batch: list[_Ticket]) -> None:
        circuits = [c for t in batch for c in t.circuits]
        values = [p for t in batch for p in t.parameter_values]
        try:
            results = self._evaluator.evaluate_circuits(circuits, values)
            self.n_batches += 1
            cur = 0
            for t in batch:
                t.result = list(results[cur : cur + len(t.circuits)])
                cur += len(t.circuits)
        except BaseException as exc:  # every caller of the merged batch sees the failure
            for t in batch:
                t.error = exc
        finally:
            for t in batch:
                t.done.set()
