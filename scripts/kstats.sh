#!/bin/bash
# rocprofv3 kernel-trace statistics of one short bench run (no counters): usage scripts/kstats.sh <tag> [env assignments..]
tag=${1:-k}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp QSV_BENCH_WINDOWS=1 QSV_BENCH_PREWARM_S=0.02
for kv in "$@"; do export "$kv"; done
out=gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $out/bench.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_us={float(r['TotalDurationNs'])/1e3:10.1f} pct={r['Percentage']}")
PY
tail -c 300 $out/bench.log | head -c 200; echo
