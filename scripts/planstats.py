"""Scheduler statistics over EVQE populations (no GPU): passes, rounds, exchanges and where gate controls end up."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import plan_interpreter as pi
from queasars_amd import workloads as helpers
from queasars_amd.planning import build_plan_words

cases = [(12, 4), (16, 4), (20, 4), (20, 8), (24, 4)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
P = 16
for n, L in cases:
    keys = ["passes", "rounds", "exchanges", "swap_rounds", "swaps", "intra", "conflicts", "gates", "u", "ctrl_global", "ctrl_wave", "ctrl_lane", "ctrl_reg", "units"]
    tot = dict.fromkeys(keys, 0.0)
    _, circuits, _ = helpers.population_circuits(n, L, P, seed=0)
    for c in circuits:
        plan = pi.decode(build_plan_words(c))
        tot["passes"] += plan["n_passes"]
        for ps in plan["passes"]:
            t = ps["t"]
            tot["rounds"] += len(ps["rounds"])
            for rd in ps["rounds"]:
                tot["swap_rounds"] += bool(rd["swaps"]); tot["swaps"] += len(rd["swaps"])
                if rd["write_cols"] is not None:
                    tot["exchanges"] += 1
                    tot["intra"] += rd["intra_wave"]
                    tot["conflicts"] += pi.bank_conflicts_b64(rd["write_cols"][:t], True) + pi.bank_conflicts_b64(rd["read_cols"][:t], False)
                for g in rd["gates"]:
                    g = dict(g, cg=g["cg"] | g["ncg"], ct=(g["ct"] | g["ct"] >> 9) & 0x1FF)  # (which index bit, set or clear)
                    tot["gates"] += 1
                    if g["creg"] is not None:
                        tot["ctrl_reg"] += 1; tot["units"] += 0.5
                    elif g["cg"]:
                        tot["ctrl_global"] += 1; tot["units"] += 0.5
                    elif g["ct"] and (g["ct"] & 63) == 0:
                        tot["ctrl_wave"] += 1; tot["units"] += 0.5
                    elif g["ct"]:
                        tot["ctrl_lane"] += 1; tot["units"] += 1.0
                    else:
                        tot["u"] += 1; tot["units"] += 1.0
    print(f"n={n} L={L}: " + "  ".join(f"{k}={tot[k] / P:.2f}" for k in keys))
