#!/bin/bash
# kernel statistics of the estimator evaluator on one population: usage scripts/evalkstats.sh n P L
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/eval_kstats_$1
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/evalprof.py "$@" > $out/run.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_us={float(r['TotalDurationNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
