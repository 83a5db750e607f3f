"""EVQE genome restatement (workload generator / input format of the hot path)."""

from queasars_amd.evqe.genome import (  # noqa: F401
    ControlGate,
    ControlledRotationGate,
    EVQECircuitLayer,
    EVQECircuitLayerException,
    EVQEGate,
    EVQEGateType,
    EVQEIndividual,
    EVQEIndividualException,
    EVQEPopulation,
    IdentityGate,
    RotationGate,
    new_random_seed,
    parameter_names,
    sorted_parameter_rank,
)
from queasars_amd.evqe.solver import (  # noqa: F401,E402
    NFT,
    SPSA,
    BestIndividualRelativeChangeTolerance,
    EVQEMinimumEigensolver,
    EVQEMinimumEigensolverConfiguration,
    EVQEResult,
    SPSATerminationChecker,
)
