"""A primitive together with the one run option the evaluators pass to it.

Mirror of ``queasars/circuit_evaluation/configured_primitives.py:9-22`` (two plain dataclasses the reference's solver
configuration carries: the sampler with its shot count, the estimator with its precision).  Here the primitive is
normally a :class:`queasars_amd.primitives.GpuSampler` / :class:`~queasars_amd.primitives.GpuEstimator`, but any object
with the same ``run`` shape is accepted, as in the reference.

``evaluator_for`` turns such a pair into the GPU evaluator of this package that does the same job, which is what a
solver configured the reference's way needs when it is pointed at this backend.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

from queasars_amd.circuit_evaluation.bitstring_evaluation import BitstringEvaluator
from queasars_amd.circuit_evaluation.circuit_evaluation import (
    BaseCircuitEvaluator,
    BitstringCircuitEvaluator,
    OperatorCircuitEvaluator,
    OperatorSamplerCircuitEvaluator,
)
from queasars_amd.ir import CircuitIR, PauliOperator


@dataclass
class ConfiguredSamplerV2:
    """:param sampler: SamplerV2-shaped primitive
    :param shots: measurements per circuit, must be positive"""

    sampler: Any
    shots: int

    def __post_init__(self) -> None:
        if int(self.shots) <= 0:
            raise ValueError("shots must be a positive integer!")


@dataclass
class ConfiguredEstimatorV2:
    """:param estimator: EstimatorV2-shaped primitive
    :param precision: target precision of the expectation values, 0 for exact"""

    estimator: Any
    precision: float

    def __post_init__(self) -> None:
        if float(self.precision) < 0:
            raise ValueError("precision must not be negative!")


def _backend_options(primitive: Any) -> dict:
    """dtype / device / seed of a Gpu* primitive; defaults for a foreign one."""
    describe = getattr(primitive, "backend_options", None)
    return dict(describe()) if callable(describe) else {}


def evaluator_for(
    configured: "ConfiguredSamplerV2 | ConfiguredEstimatorV2",
    operator: Optional[PauliOperator] = None,
    bitstring_evaluator: Optional[BitstringEvaluator] = None,
    alpha: float = 1.0,
    initial_state_circuit: Optional[CircuitIR] = None,
) -> BaseCircuitEvaluator:
    """The evaluator the reference's solvers build from their configuration
    (``evolving_ansatz_minimum_eigensolver.py`` picks OperatorCircuitEvaluator for a configured estimator,
    OperatorSamplerCircuitEvaluator for a sampler + operator, BitstringCircuitEvaluator for a sampler + bitstring
    evaluator), on the GPU backend."""
    if (operator is None) == (bitstring_evaluator is None):
        raise ValueError("Exactly one of operator and bitstring_evaluator must be given!")
    if isinstance(configured, ConfiguredEstimatorV2):
        if operator is None:
            raise ValueError("An estimator can only evaluate an operator!")
        return OperatorCircuitEvaluator(
            operator,
            estimator_precision=float(configured.precision),
            initial_state_circuit=initial_state_circuit,
            **_backend_options(configured.estimator),
        )
    if isinstance(configured, ConfiguredSamplerV2):
        options = _backend_options(configured.sampler)
        if operator is not None:
            return OperatorSamplerCircuitEvaluator(
                int(configured.shots), operator, alpha=alpha, initial_state_circuit=initial_state_circuit, **options
            )
        return BitstringCircuitEvaluator(
            int(configured.shots),
            bitstring_evaluator,
            alpha=alpha,
            initial_state_circuit=initial_state_circuit,
            **options,
        )
    raise TypeError("configured must be a ConfiguredSamplerV2 or a ConfiguredEstimatorV2")
