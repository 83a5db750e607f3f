"""config 5's genome through the 2^n state, later-pass kernel only (bench.config5_sweep_block's through_the_state rows): n precision"""
import sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
rows = bench.config5_sweep_block(tuple(int(a) for a in sys.argv[1:]) or (28,))
for k, row in rows.items():
    if k == "note":
        continue
    for prec in ("fp64", "fp32"):
        e = row[prec]["through_the_state"]
        print(k, prec, "later pass", round(e.get("later_pass_launch_us", 0), 1), "us", round(e.get("later_pass_frac_hbm", 0), 3), "of 8 TB/s;", round(e["evals_per_s"], 2), "evals/s; default route", round(row[prec]["default_route"]["evals_per_s"], 1))
