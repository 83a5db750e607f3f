"""Ablation builds of libqsv (timing experiments; results are wrong by construction): each variant switches one part of
the gate-pass kernel off so that its cost in the whole launch can be read from the per-kernel launch times.

    python scripts/ablate.py build                 # here: queasars_amd/libqsv_abl_<name>.so for every variant
    python scripts/ablate.py run [n P]             # on the GPU box: one line per variant
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
VARIANTS = {
    "base": (),
    "nogates": ("QSV_ABL_NOGATES",),
    "noswap": ("QSV_ABL_NOSWAP",),
    "noload": ("QSV_ABL_NOLOAD",),
    "nof": ("QSV_ABL_NOF",),
    "nodiag": ("QSV_ABL_NODIAG",),
    "noload_nof_nodiag": ("QSV_ABL_NOLOAD", "QSV_ABL_NOF", "QSV_ABL_NODIAG"),
    "nothing_but_memory": ("QSV_ABL_NOGATES", "QSV_ABL_NOSWAP"),
    "nothing_but_gates": ("QSV_ABL_NOLOAD", "QSV_ABL_NOF", "QSV_ABL_NODIAG", "QSV_ABL_NOSWAP"),
    # the contraction kernel of split evaluations
    "ct_nod": ("QSV_ABL_CT_NOD",),
    "ct_notab": ("QSV_ABL_CT_NOTAB",),
    "ct_nomath": ("QSV_ABL_CT_NOMATH",),
    "ct_nod_notab": ("QSV_ABL_CT_NOD", "QSV_ABL_CT_NOTAB"),
    "ct_nothing": ("QSV_ABL_CT_NOD", "QSV_ABL_CT_NOTAB", "QSV_ABL_CT_NOMATH"),
}
# variants of the GENERATED assembly round loop (production path): parts left out by gen_gate_loop.py (QSV_GEN_ABL)
ASM_VARIANTS = {
    "asm_gatevalu": "gatevalu",
    "asm_pairtest": "pairtest",
    "asm_swapvalu": "swapvalu",
    "asm_gateloop": "gateloop",
    "asm_gatevalu_swapvalu": "gatevalu,swapvalu",
    "asm_gateloop_swapvalu": "gateloop,swapvalu",
    # the skeleton: no gates, no swaps, no memory -- what the launch, the set-up and the loop structure cost by themselves
    "asm_skeleton": ("gateloop,swapvalu", ("QSV_ABL_NOLOAD", "QSV_ABL_NOF", "QSV_ABL_NODIAG")),
    "asm_nomem": ("", ("QSV_ABL_NOLOAD", "QSV_ABL_NOF", "QSV_ABL_NODIAG")),
}


def lib_of(name):
    return ROOT / "queasars_amd" / f"libqsv_abl_{name}.so"


def main():
    if sys.argv[1] == "build":
        from queasars_amd import _build

        only = set(sys.argv[2:])
        for name, defines in VARIANTS.items():
            if name == "base" or (only and name not in only):
                continue
            print(_build.build(force=True, defines=defines, lib_path=lib_of(name)))
        csrc = ROOT / "queasars_amd" / "csrc"
        for name, abl in ASM_VARIANTS.items():
            if only and name not in only:
                continue
            abl, extra = abl if isinstance(abl, tuple) else (abl, ())
            inc = csrc / f"gate_loop_{name}.inc"
            subprocess.run([sys.executable, str(csrc / "gen_gate_loop.py"), "--out", str(inc)],
                           env=dict(os.environ, QSV_GEN_ABL=abl), check=True)
            print(_build.build(force=True, defines=(f'QSV_GATE_LOOP_INC="{inc.name}"', *extra), lib_path=lib_of(name)))
            inc.unlink()
        return
    n, pop = (sys.argv[2:4] + ["20", "64"][len(sys.argv[2:4]):])
    names = [v for v in list(VARIANTS) + list(ASM_VARIANTS) if v == "base" or lib_of(v).exists()]
    for name in names:
        env = dict(os.environ, QSV_BENCH_QUBITS=n, QSV_BENCH_POP=pop)
        if name != "base":
            env["QSV_LIBRARY"] = str(lib_of(name))
        res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"],
                             env=env, capture_output=True, text=True)
        if res.returncode != 0:
            print(name, "FAILED", res.stderr[-400:])
            continue
        d = json.loads(res.stdout.strip().splitlines()[-1])
        print(f"{name:22s} {d['value']:10.0f} evals/s  kernels us: " + "  ".join(f"{k['avg_launch_us']:.1f}" for k in d["roofline"]["kernels"]), flush=True)


if __name__ == "__main__":
    main()
