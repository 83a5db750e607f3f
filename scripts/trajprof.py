"""Where a realistic EVQE run's wall clock goes (bench.py's trajectory block under cProfile, on the GPU).
  python scripts/trajprof.py [generations] [top]"""
import cProfile, pstats, sys, time, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 8
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
op = bench.ising_operator(bench.N_QUBITS, 2020)
bench.trajectory_block(op, 2)  # (code objects, library, clocks)
prof = cProfile.Profile()
prof.enable()
out = bench.trajectory_block(op, gens)
prof.disable()
print(round(out["evals_per_s"]), "evals/s", out["evaluations"], "evaluations", round(out["seconds"], 3), "s")
for g in out["generations"]:
    print("  gen", g["generation"], g["evaluations"], "evals", round(g["seconds"] * 1e3, 1), "ms", round(g["evals_per_s"]), "evals/s", g["routes"])
pstats.Stats(prof).sort_stats("cumulative").print_stats(top)
pstats.Stats(prof).sort_stats("tottime").print_stats(25)
