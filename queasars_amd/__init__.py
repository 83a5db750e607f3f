"""MI355X-native statevector backend for QUEASARS' circuit-evaluation path."""

import os as _os

# A handle owns three or four HIP streams (two lanes for a batch's pushes, one for the unsplit circuits of a mixed batch), and
# the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- four by default: with two handles alive
# (two evaluators, one per operator) streams share queues and chains of launches that should run side by side run one after
# the other (config 3's step next to a second handle: 0.34 ms against 0.22 ms).  The runtime reads the variable when it
# initialises, i.e. at the process's first HIP call: setting it here works as long as that has not happened yet, and never
# overrides what the user has set.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from queasars_amd.ir import CircuitIR, ParamRef, PauliOperator  # noqa: F401

__version__ = "0.1.0"
