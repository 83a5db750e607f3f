"""One parameter search of a whole population (EVQE's last-layer search: one SPSA run per individual, all advanced in
lock-step, every iteration ONE evaluate_circuits call of two points per individual): wall time with the whole-array driver of
evqe/solver.py and with the run-by-run one (QSV_SCALAR_SPSA=1).  usage: spsatime.py [n L P]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from queasars_amd import workloads
from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
from queasars_amd.evqe import EVQEPopulation
from queasars_amd.evqe import solver as S

n, L, P = (int(x) for x in (sys.argv[1:4] + ["20", "4", "64"][len(sys.argv[1:4]):]))
pop = EVQEPopulation.random_population(n, L, P, True, 0)
ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, seed=2020))
cfg = S.SPSA()  # (the notebooks' configuration: 33 iterations, two evaluations each)


def jobs():
    out = []
    for k, ind in enumerate(pop.individuals):
        circuit = ind.get_partially_parameterized_quantum_circuit({-1})
        out.append((circuit, cfg.new_run(ind.get_layer_parameter_values(-1), seed=k)))
    return out


if len(sys.argv) > 4 and sys.argv[4] == "checker":
    cfg = S.SPSA(termination_checker=S.SPSATerminationChecker(0.01, 2))
reference_x = None
for mode in ("on device", "vectorised", "run by run"):
    if mode == "run by run":
        os.environ["QSV_SCALAR_SPSA"] = "1"
    S._minimize_batched(ev, jobs(), on_device=mode == "on device")  # (registers the circuits, warms everything)
    t = []
    for _ in range(5):
        j = jobs()
        t0 = time.perf_counter()
        S._minimize_batched(ev, j, on_device=mode == "on device")
        t.append(time.perf_counter() - t0)
    evals = sum(run.nfev for _, run in j)
    best = min(t)
    xs = np.concatenate([run.x for _, run in j])
    if mode == "vectorised":
        reference_x = xs
    note = "" if mode != "run by run" else f"; max |x - vectorised x| = {np.abs(xs - reference_x).max():.1e}"
    if mode == "on device":
        device_x = xs
    print(f"{mode:11s}: {best * 1e3:7.2f} ms per search of {P} individuals ({evals} evaluations, {best / cfg.maxiter * 1e6:6.1f} us per "
          f"iteration, {evals / best:9.0f} evaluations per second); x[0][:3] = {j[0][1].x[:3]}{note}")
print(f"max |x on device - x vectorised| = {np.abs(device_x - reference_x).max():.2e}")
