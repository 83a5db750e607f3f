#!/bin/bash
# rocprofv3 runs beside the headline profile: the single-gate microbenchmark at n = 24 and 26 (kernel trace + stats, then
# FETCH_SIZE and WRITE_SIZE in PMC passes of their own) and BASELINE config 5 (general 500-term operator at n = 28).
# usage: scripts/profile_extra.sh <tag>
set -u
tag=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_extra_$tag
rm -rf $out; mkdir -p $out
for n in 24 26; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/micro${n}_trace -- python3 scripts/microbench.py --n $n --reps 20 --targets 0,3,7,11,12,17,21,23 > $out/micro${n}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/micro${n}_$c -- python3 scripts/microbench.py --n $n --reps 5 --targets 0,3,7,11,12,17,21,23 > $out/micro${n}_$c.log 2>&1
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/config5_trace -- python3 scripts/configs.py --configs 5 > $out/config5_trace.log 2>&1
python3 - "$out" <<'PY' > $out/summary.txt 2>&1
import csv, glob, sys, json
root = sys.argv[1]
for n in (24, 26):
    print(f"== microbenchmark n = {n}: one u / cu3 gate per read-modify-write sweep of a 2^{n} fp64 state ({16 << n >> 20} MiB) ==")
    for path in glob.glob(f"{root}/micro{n}_trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pass_kernel" in r["Name"]:
                avg = float(r["AverageNs"])
                print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_ns={avg:10.0f}  -> {32.0 * (1 << n) / avg:8.1f} GB/s algorithmic (32 * 2^n bytes per sweep)")
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        xs = []
        for path in glob.glob(f"{root}/micro{n}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                if "pass_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    xs.append(float(r["Counter_Value"]))
        if xs:
            vals[c] = sum(xs) / len(xs)
            print(f"  {c} mean per sweep = {vals[c]:.1f} KiB over {len(xs)} dispatches")
    if len(vals) == 2:
        hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
        print(f"  HBM-side bytes per sweep (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) = {hbm / 2**20:.1f} MiB; algorithmic = {32 * (1 << n) / 2**20:.0f} MiB")
print("== BASELINE config 5 (n = 28, 500 Pauli strings, fp64 then fp32) ==")
for path in glob.glob(f"{root}/config5_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>5s} total_ns={r['TotalDurationNs']:>12s} avg_ns={float(r['AverageNs']):12.0f} pct={r['Percentage']}")
for line in open(f"{root}/config5_trace.log"):
    if line.startswith("{"):
        print("  " + line.strip())
PY
cat $out/summary.txt
