#!/bin/bash
# rocprofv3 kernel-trace statistics of any python command: usage scripts/kstats_cmd.sh <tag> <script> [args..]
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/run.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_us={float(r['TotalDurationNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
rm -rf $out/trace
