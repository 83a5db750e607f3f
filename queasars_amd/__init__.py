"""MI355X-native statevector backend for QUEASARS' circuit-evaluation path."""

# (Hardware queues: a handle owns three or four HIP streams and the HIP runtime maps a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues, four by default.  Two handles alive at once share queues -- chains of launches that should
# run side by side then run one after the other: 0.34 ms against 0.22 ms for config 3's step -- and GPU_MAX_HW_QUEUES=8 in the
# process's environment cures that; but it also puts RCCL's internal stream on a queue of its own, and the chained all-gather
# step of a sharded population then pays for cross-queue waits: 111 -> 170-185 us.  The package therefore leaves the runtime's
# default alone)

from queasars_amd.ir import CircuitIR, ParamRef, PauliOperator  # noqa: F401

__version__ = "0.1.0"
