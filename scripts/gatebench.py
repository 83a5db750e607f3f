"""Steady-state cost model of the pass kernel: time one read-modify-write sweep of a 2^n state for circuits that
fit one pass, as a function of the number of gates and of which tile bits they hit (rounds / exchanges)."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from queasars_amd.circuit_evaluation import StatevectorDevice  # noqa: E402
from queasars_amd.ir import CircuitIR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    n = args.n
    dev = StatevectorDevice(n, group=1)
    sweep_bytes = 32 * (1 << n)

    def run(name, circuit):
        ms, passes = dev.bench_ops(circuit, args.reps)
        print(json.dumps({"case": name, "gates": len(circuit), "passes": passes, "ms": round(ms, 4),
                          "GBps_per_pass": round(passes * sweep_bytes / ms / 1e6, 0)}), flush=True)

    hi = list(range(n - 4, n))       # 4 high qubits: default register bits, no exchange
    mid = list(range(8, 12))
    low = list(range(0, 4))
    run("empty (id only)", CircuitIR(n).id(0))
    for reps in (1, 2, 4, 8):
        c = CircuitIR(n)
        for r in range(reps):
            for q in hi:
                c.u(0.3 + r, 0.2, 0.1, q)
        run(f"{4*reps} u on 4 high qubits (1 round)", c)
    c = CircuitIR(n)
    for q in hi + mid:
        c.u(0.3, 0.2, 0.1, q)
    run("8 u: high + mid (2 rounds)", c)
    c = CircuitIR(n)
    for q in hi + mid + low:
        c.u(0.3, 0.2, 0.1, q)
    run("12 u: high + mid + low (3-4 rounds)", c)
    c = CircuitIR(n)
    for q in low:
        c.u(0.3, 0.2, 0.1, q)
    run("4 u on low qubits (exchange in + out)", c)
    c = CircuitIR(n)
    for r in range(3):
        for q in hi + mid + low:
            c.u(0.3 + r, 0.2, 0.1, q)
    run("36 u: 3 x (high + mid + low)", c)
    c = CircuitIR(n)
    for i, q in enumerate(hi):
        c.cu3(0.3, 0.2, 0.1, (q + 5) % (n - 4), q)
    run("4 cu3 high targets, various controls", c)


if __name__ == "__main__":
    main()
