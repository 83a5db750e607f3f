#!/bin/bash
# usage: timeline_env.sh <tag> "ENV=.. ENV=.."   -- kernel trace of a short bench run under the given environment, last step's timeline
set -u
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp QSV_BENCH_WINDOWS=1 QSV_BENCH_PREWARM_S=0.02
for kv in $1; do export "$kv"; done
out=gpurun_out/tl_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $out/bench.log 2>&1
python3 scripts/timeline.py $out/trace ${GAP_US:-12} ${LAST:-} > $out/timeline.txt 2>&1
echo "== $tag: $1"; tail -1 $out/bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'evals/s', d['ms_per_step'])"
cat $out/timeline.txt
rm -rf $out/trace
