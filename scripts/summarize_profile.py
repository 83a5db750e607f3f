"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a short text summary."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name[:70]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for path in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    with open(path) as f:
        rows = list(csv.DictReader(f))
    for r in rows:
        print(f"{short(r['Name']):72s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):>10.0f} pct={r['Percentage']}")

print("\n== per-kernel PMC averages per dispatch ==")
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(f"{root}/pmc_*/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, counters in acc.items():
    if not any(k in kern for k in ("pass_kernel", "contract_kernel", "factor_", "reduce", "prepare")):
        continue
    print(kern)
    for cname, vals in sorted(counters.items()):
        print(f"    {cname:28s} n={len(vals):5d} mean={sum(vals)/len(vals):16.1f}")

# The gate-pass kernel has two instantiations (FIRST = true: the synthesising pass 0; false: every later pass).
# bench.py times "a pass launch" over both, so the summary gives the combined figures next to the per-kernel ones.
import json
stats = {}
for path in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "pass_kernel" in r["Name"] or "contract_kernel" in r["Name"] or "factor_" in r["Name"]:
                stats[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
if stats:
    calls = sum(c for c, _ in stats.values())
    total = sum(t for _, t in stats.values())
    print("\n== pass_kernel (both instantiations) + contract_kernel / factor kernels ==")
    print(f"calls={calls} total_ns={total:.0f} avg_ns={total / calls:.0f}")

# HBM traffic per launch and per instantiation of the gate-pass kernel, corrected as MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950; WRITE_SIZE is exact; both in KiB.
# bench.py reads this file (profiles/traffic.json) for roofline.traffic.
kernels = {}
for kern, counters in acc.items():
    if "pass_kernel" not in kern and "contract_kernel" not in kern and "factor_" not in kern:
        continue
    kind = "2" if ("contract_kernel" in kern or "factor_" in kern) else ("0" if ", true>" in kern else "1")
    fetch, write = counters.get("FETCH_SIZE", []), counters.get("WRITE_SIZE", [])
    if fetch and write:
        f_mean, w_mean = sum(fetch) / len(fetch), sum(write) / len(write)
        if kind in kernels:  # (the two factor kernels run as a pair, which bench.py times as one: their means add)
            k = kernels[kind]
            k["kernel"] += " + " + kern
            k["FETCH_SIZE_KiB_mean"] += f_mean
            k["WRITE_SIZE_KiB_mean"] += w_mean
            k["hbm_bytes_per_launch"] += (2.0 * f_mean + w_mean) * 1024.0
            continue
        kernels[kind] = {
            "kernel": kern,
            "FETCH_SIZE_KiB_mean": f_mean,
            "WRITE_SIZE_KiB_mean": w_mean,
            "hbm_bytes_per_launch": (2.0 * f_mean + w_mean) * 1024.0,
            "dispatches_sampled": len(fetch),
        }
if kernels:
    out = {
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --steps 3 --warmup 1 --no-extras "
        f"--no-cpu-baseline` ({root}); reads = 2 x FETCH_SIZE (gfx950 correction), writes = WRITE_SIZE, KiB -> bytes",
        "kernels": kernels,
    }
    print("\n== traffic ==")
    print(json.dumps(out))
    with open(f"{root}/traffic.json", "w") as f:
        json.dump(out, f, indent=1)
