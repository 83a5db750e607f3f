"""Per-step wall time of the cold path: every step evaluates 64 structures the device has never seen."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation

n, L, P = (int(x) for x in (sys.argv[1:4] + ["20", "4", "64"][len(sys.argv[1:4]):]))
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
dev = ev.statevector_device
pops = []
for s in range(5):
    pop = EVQEPopulation.random_population(n, L, P, True, 1000 + s)
    pops.append(([i.get_parameterized_quantum_circuit() for i in pop.individuals], [list(i.parameter_values) for i in pop.individuals]))
# (the library call inside _register_many, timed on its own)
inner = {"t": 0.0}
real_create = dev._lib.qsv_circuits_create


def timed_create(*args):
    t = time.perf_counter()
    rc = real_create(*args)
    inner["t"] = time.perf_counter() - t
    return rc


class LibProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        return timed_create if name == "qsv_circuits_create" else getattr(self._lib, name)


dev._lib = LibProxy(dev._lib)
for k, (cs, ps) in enumerate(pops):
    t0 = time.perf_counter()
    fresh = [c for c in cs if dev._serial not in c._registered]
    dev._register_many(fresh)
    t1 = time.perf_counter()
    ev.evaluate_circuits(cs, ps)
    t2 = time.perf_counter()
    ev.evaluate_circuits(cs, ps)
    t3 = time.perf_counter()
    print(f"step {k}: register {1e6 * (t1 - t0):8.0f} us (library call {1e6 * inner['t']:6.0f})   first evaluation {1e6 * (t2 - t1):8.0f} us   warm evaluation {1e6 * (t3 - t2):8.0f} us")
