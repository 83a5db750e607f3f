"""Split evaluations with four and five cut keys alone: wall time per call and per-kind kernel times for subsets of an
L-layer population's circuits that need them.  usage: k5time.py [n L] [counts..]   (QSV_LIBRARY=stamped build: phases too)"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from queasars_amd import _lib, workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.ir import QSV_OP_DTYPE


def split_keys(circuit, max_side):
    ops = circuit.packed()
    cap = 4 * len(ops) + 64
    a, b = np.zeros(cap, dtype=QSV_OP_DTYPE), np.zeros(cap, dtype=QSV_OP_DTYPE)
    na, nb, mask = C.c_int(0), C.c_int(0), C.c_uint64(0)
    return _lib.load().qsv_split_describe(circuit.n_qubits, len(ops), _lib.as_ptr(ops), max_side, C.byref(mask), _lib.as_ptr(a), cap,
                                          C.byref(na), _lib.as_ptr(b), cap, C.byref(nb))


n, L = (int(x) for x in (sys.argv[1:3] + ["20", "6"][len(sys.argv[1:3]):]))
counts = [int(x) for x in sys.argv[3:]] or [1, 2, 8, 26]
_, circuits, params = workloads.population_circuits(n, L, 64, seed=0)
limit = 16 if n < 21 else 17
keys = [split_keys(c, limit) for c in circuits]
big = [i for i, k in enumerate(keys) if k >= 4]
print(f"n={n} L={L}: keys {sorted(keys)}; {len(big)} circuits with four or five keys")
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
dev = ev.statevector_device
dev.set_option("split_max_keys", 5)
for count in counts:
    idx = big[:count]
    cs, ps = [circuits[i] for i in idx], [params[i] for i in idx]
    for _ in range(20):
        ev.evaluate_circuits(cs, ps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        ev.evaluate_circuits(cs, ps)
    dt = (time.perf_counter() - t0) / reps
    dev.set_profiling(True)
    ev.evaluate_circuits(cs, ps)
    prof = dev.profile()
    dev.set_profiling(False)
    kinds = [f"{prof['kernel_ms'][k] * 1e3:.1f}us/{int(prof['kernel_launches'][k])}" for k in range(3)]
    print(f"  {len(idx):3d} circuits (keys {[keys[i] for i in idx][:8]}..): {dt * 1e6:7.1f} us per call   kernels [virtual circuits, later passes, factor pair]: {kinds}", flush=True)
    if os.environ.get("QSV_LIBRARY"):  # a stamped build: mean shader cycles per wave by pass and phase, this subset alone
        PH = ["setup", "load/synth", "p:hdr", "p:params", "p:mats", "p:factors", "p:tables", "p:tileinfo", "x", "x", "gates", "store/red", "epilogue", "prep-rest"]
        table = (C.c_ulonglong * 128)()
        lib = _lib.load()
        lib.qsv_debug_stamps(table, 1)
        for _ in range(10):
            ev.evaluate_circuits(cs, ps)
        lib.qsv_debug_stamps(table, 1)
        t = np.asarray(list(table), dtype=np.float64).reshape(8, 16)
        for p in range(8):
            if t[p, 15]:
                per = t[p, :14] / t[p, 15]
                print(f"    pass {p}: waves/call {t[p, 15] / 10:.0f}  " + "  ".join(f"{name} {c:.0f}" for name, c in zip(PH, per) if c >= 1) + f"  total {per.sum():.0f}")
