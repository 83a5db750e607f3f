#!/bin/bash
# usage: scripts/sweep.sh "<env assignments>" ...   -> one bench line per configuration
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg" >> gpurun_out/sweep.log
  env $cfg timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']
        print('value=%.0f evals/s ms_per_step=%.3f launches=%d avg_launch_ms=%.4f achieved=%.0f GB/s frac=%.3f' % (d['value'], d['ms_per_step'], r['launches'], r['avg_launch_ms'], r['achieved'], r['frac']))
" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
