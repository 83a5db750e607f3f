"""What each link of the one-launch route's chain costs: the launch timed with that link left out (results wrong by
construction), one zero-key circuit alone and the benchmark population (n = 20, P = 64, L = 4).

    python scripts/chain_ablation.py build        # here: queasars_amd/libqsv_chain_<name>.so
    python scripts/chain_ablation.py run          # on the GPU box
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
DEFINES = {
    "no_sincos": ("QSV_ABL_PREP_TRIG",),
    "no_tables": ("QSV_ABL_PREP_TABLES",),
    "no_gram": ("QSV_ABL_TAIL_GRAM",),
    "no_pair_steps": ("QSV_ABL_PAIR_STEPS",),
    "pair_steps_without_diagonals": ("QSV_ABL_PAIR_NODIAG",),
    "no_handoff_no_combination": ("QSV_ABL_NO_HANDOFF",),
    "preparation_only": ("QSV_ABL_AFTER_PREP",),
    "empty_kernel": ("QSV_ABL_EMPTY",),
    "plain_stores": ("QSV_ABL_NO_THROUGH",),
}
ASM = {"no_gates": "gateloop", "no_swaps": "swapvalu", "no_gates_no_swaps": "gateloop,swapvalu"}


def lib_of(name):
    return ROOT / "queasars_amd" / f"libqsv_chain_{name}.so"


def main():
    if sys.argv[1] == "build":
        from queasars_amd import _build

        for name, defines in DEFINES.items():
            print(_build.build(force=True, defines=defines, lib_path=lib_of(name)))
        csrc = ROOT / "queasars_amd" / "csrc"
        for name, abl in ASM.items():
            inc = csrc / f"gate_loop_{name}.inc"
            subprocess.run([sys.executable, str(csrc / "gen_gate_loop.py"), "--out", str(inc)], env=dict(os.environ, QSV_GEN_ABL=abl), check=True)
            print(_build.build(force=True, defines=(f'QSV_GATE_LOOP_INC="{inc.name}"',), lib_path=lib_of(name)))
            inc.unlink()
        return
    if sys.argv[1] == "one":  # (a child process: one library)
        import time
        import numpy as np
        from queasars_amd import workloads
        from queasars_amd.circuit_evaluation import OperatorCircuitEvaluator
        from queasars_amd.evqe import EVQEPopulation

        n, P, L = 20, 64, 4
        pop = EVQEPopulation.random_population(n, L, P, True, 0)
        circuits = [ind.get_parameterized_quantum_circuit() for ind in pop.individuals]
        params = [list(ind.parameter_values) for ind in pop.individuals]
        ev = OperatorCircuitEvaluator(workloads.random_ising_operator(n, 2020))
        dev = ev.statevector_device

        def launch_us(idx, reps=40):
            cs, ps = [circuits[i] for i in idx], [params[i] for i in idx]
            for _ in range(5):
                ev.evaluate_circuits(cs, ps)
            dev.set_profiling(True)
            acc, launches = 0.0, 0
            for _ in range(reps):
                ev.evaluate_circuits(cs, ps)
                p = dev.profile()
                acc += sum(p["kernel_ms"])
                launches += 1
            dev.set_profiling(False)
            return acc / launches * 1e3

        if os.environ.get("QSV_CHAIN_CIRCUITS"):  # (other circuits of the population, alone: indices)
            picked = [int(x) for x in os.environ["QSV_CHAIN_CIRCUITS"].split(",")]
            print(f"{sys.argv[2]:20s} " + "   ".join(f"circuit {i} alone {launch_us([i]):6.1f} us" for i in picked), flush=True)
            return
        print(f"{sys.argv[2]:20s} circuit 8 alone {launch_us([8]):6.1f} us   circuit 41 alone {launch_us([41]):6.1f} us   all 64 {launch_us(list(range(P))):6.1f} us", flush=True)
        return
    for name in ["base"] + [v for v in list(DEFINES) + list(ASM) if lib_of(v).exists()]:
        env = dict(os.environ)
        if name != "base":
            env["QSV_LIBRARY"] = str(lib_of(name))
        subprocess.run([sys.executable, __file__, "one", name], env=env)


if __name__ == "__main__":
    main()
