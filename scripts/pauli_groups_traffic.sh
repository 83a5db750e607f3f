#!/bin/bash
# pauli_groups_kernel at n = 28, 500 Pauli strings, through the 4 GiB state (QSV_FACTOR=0): kernel time and FETCH_SIZE.
# usage: scripts/pauli_groups_traffic.sh  -> gpurun_out/pauli_groups/summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp QSV_FACTOR=0
out=gpurun_out/pauli_groups
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/configs.py --configs 5 > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 scripts/configs.py --configs 5 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- python3 scripts/configs.py --configs 5 > $out/tcc.log 2>&1
python3 - "$out" <<'PY' > $out/summary.txt 2>&1
import csv, glob, sys
root = sys.argv[1]
print("pauli_groups_kernel, n = 28, 500 random Pauli strings over {I,X,Y,Z} (default_rng(2028)), state path (QSV_FACTOR=0)")
for path in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "pauli" in r["Name"] or "pass_kernel" in r["Name"]:
            print(f"  {r['Name'][:64]:64s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs']) / 1e6:10.3f}")
for name, cs in (("fetch", ("FETCH_SIZE",)), ("tcc", ("TCC_HIT_sum", "TCC_MISS_sum"))):
    acc = {}
    for path in glob.glob(f"{root}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pauli_groups" in r["Kernel_Name"]:
                acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"  {k:40s} {c:14s} n={len(v)} mean={sum(v) / len(v):.1f}")
PY
cat $out/summary.txt; grep "^{" $out/trace.log
