"""GPU-backed mirror of ``queasars.circuit_evaluation``."""

from queasars_amd.circuit_evaluation.bitstring_evaluation import BitstringEvaluator, BitstringEvaluatorException  # noqa: F401
from queasars_amd.circuit_evaluation.circuit_evaluation import (  # noqa: F401
    BaseCircuitEvaluator,
    BitstringCircuitEvaluator,
    CircuitEvaluatorException,
    KeptState,
    OperatorCircuitEvaluator,
    OperatorSamplerCircuitEvaluator,
    StatevectorDevice,
    measure_quasi_distributions,
)
from queasars_amd.circuit_evaluation.configured_primitives import (  # noqa: F401
    ConfiguredEstimatorV2,
    ConfiguredSamplerV2,
    evaluator_for,
)
from queasars_amd.circuit_evaluation.coalescing import CoalescingCircuitEvaluator  # noqa: F401
from queasars_amd.circuit_evaluation.expectation_calculation import (  # noqa: F401
    get_expectation_with_bitstring_evaluator,
    get_expectation_with_operator,
)
