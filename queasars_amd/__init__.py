"""MI355X-native statevector backend for QUEASARS' circuit-evaluation path."""

from queasars_amd.ir import CircuitIR, ParamRef, PauliOperator  # noqa: F401

__version__ = "0.1.0"
