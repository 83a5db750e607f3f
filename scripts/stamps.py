"""Per-phase shader cycles of pass_kernel, from a diagnostic build (-DQSV_STAMPS) of the same sources.

    python scripts/stamps.py build          # here (no GPU needed): writes queasars_amd/libqsv_stamps.so
    python scripts/stamps.py run [n P L]    # on the GPU box

Every wave stamps s_memtime (after draining vmcnt, so memory waits are charged to the phase that
issued the loads) at the phase boundaries; the table is [pass][phase] summed over workgroups.
"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
STAMP_LIB = Path(os.environ.get("QSV_STAMP_LIB", ROOT / "queasars_amd" / "libqsv_stamps.so"))
PHASES = ["setup", "load/synth", "x-wait", "x-wr-re", "x-bar1", "x-rd-re", "x-bar2", "x-wr-im", "x-bar3", "x-rd-im", "gates",
          "store/red", "epilogue", "prepare"]


def main() -> None:
    if sys.argv[1] == "build":
        from queasars_amd import _build

        # "build asm": stamps around the production assembly round loop (all of it charged to "gates")
        defines = ("QSV_STAMPS", "QSV_STAMPS_ASM") if "asm" in sys.argv[2:] else ("QSV_STAMPS",)
        if "wave0" in sys.argv[2:]:  # only the first wave of every workgroup is counted
            defines += ("QSV_STAMPS_WAVE0",)
        print(_build.build(force=True, defines=defines, lib_path=STAMP_LIB))
        return
    os.environ["QSV_LIBRARY"] = str(STAMP_LIB)
    import numpy as np
    from queasars_amd import workloads as helpers
    from queasars_amd import _lib
    from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator

    n, P, L = (int(x) for x in (sys.argv[2:5] + ["20", "64", "4"][len(sys.argv[2:5]):]))
    _, circuits, params = helpers.population_circuits(n, L, P, seed=0)
    if os.environ.get("QSV_STAMP_ONLY"):  # (some circuits of the population alone: indices)
        only = [int(x) for x in os.environ["QSV_STAMP_ONLY"].split(",")]
        circuits, params = [circuits[i] for i in only], [params[i] for i in only]
    ev = OperatorCircuitEvaluator(helpers.random_ising_operator(n, seed=2020))
    lib = _lib.load()
    table = (C.c_ulonglong * 128)()
    for _ in range(3):
        ev.evaluate_circuits(circuits, params)
    assert lib.qsv_debug_stamps(table, 1) == 0, "not a QSV_STAMPS build"
    reps = 10
    import time

    t0 = time.perf_counter()
    for _ in range(reps):
        ev.evaluate_circuits(circuits, params)
    print(f"stamped build: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per population evaluation")
    assert lib.qsv_debug_stamps(table, 1) == 0
    t = np.asarray(list(table), dtype=np.float64).reshape(8, 16)
    NP = len(PHASES)
    print(f"n={n} P={P} L={L}: mean shader cycles per wave, by pass and phase")
    print("pass  waves/batch " + " ".join(f"{p:>10s}" for p in PHASES) + "         total")
    grand = np.zeros(NP)
    for p in range(8):
        wgs = t[p, 15]
        if wgs == 0:
            continue
        per = t[p, :NP] / wgs
        if p == 7:  # (the tail of the one-launch route of split evaluations: its own phases)
            names = ["stores drained", "Gram matrices", "hand-off", "combination (second side only)"]
            print(f"tail  {wgs / reps:11.0f}   " + "   ".join(f"{nm} {c:.0f}" for nm, c in zip(names, per)) + f"   total {per[:8].sum():.0f}")
            # (inside the Gram phase, sides of up to four product terms: stamped separately, the phase's own column holds the rest)
            print("      Gram phase: " + "   ".join(f"{nm} {c:.0f}" for nm, c in zip(["rows staged (eight terms: all that is not the steps)", "blocks (eight terms: the tail of the last)", "sums across lanes", "eight terms: the blocks' steps"], per[4:8])))
            continue
        grand += t[p, :NP]
        print(f"{p:4d}  {wgs / reps:11.0f} " + " ".join(f"{c:10.0f}" for c in per) + f" {per.sum():10.0f}")
    print("share of all stamped cycles: " + "  ".join(f"{ph} {100 * g / grand.sum():.1f}%" for ph, g in zip(PHASES, grand)))


if __name__ == "__main__":
    main()
