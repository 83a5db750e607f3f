"""Roofline microbenchmark: one u / cu3 gate per sweep of a 2^n state, every position."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from queasars_amd.circuit_evaluation import StatevectorDevice  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--dtype", default="fp64")
    ap.add_argument("--tile-bits", type=int, default=0)
    ap.add_argument("--reg-bits", type=int, default=0)
    ap.add_argument("--low-bits", type=int, default=0)
    ap.add_argument("--targets", default="all")
    args = ap.parse_args()
    dev = StatevectorDevice(args.n, dtype=args.dtype, tile_bits=args.tile_bits, reg_bits=args.reg_bits, low_bits=args.low_bits, group=1)
    amp_bytes = 16 if args.dtype == "fp64" else 8
    sweep_bytes = 2 * amp_bytes * (1 << args.n)
    targets = range(args.n) if args.targets == "all" else [int(t) for t in args.targets.split(",")]
    rows = []
    for t in targets:
        ms_u = dev.bench_gate(t, -1, reps=args.reps)
        c = (t + args.n // 2) % args.n
        ms_c = dev.bench_gate(t, c, reps=args.reps)
        rows.append({"target": t, "u_ms": ms_u, "u_GBps": sweep_bytes / ms_u / 1e6, "cu3_control": c, "cu3_ms": ms_c, "cu3_GBps": sweep_bytes / ms_c / 1e6})
        print(json.dumps(rows[-1]), flush=True)
    gb = [r["u_GBps"] for r in rows] + [r["cu3_GBps"] for r in rows]
    print(json.dumps({"n": args.n, "dtype": args.dtype, "min_GBps": min(gb), "mean_GBps": sum(gb) / len(gb), "max_GBps": max(gb), "frac_of_8TBps_mean": sum(gb) / len(gb) / 8000.0}))


if __name__ == "__main__":
    main()
