#!/bin/bash
# same-box A/B of the headline against an older tree checked out (and built) under _old/: alternating runs
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for tree in _old .; do
    ( cd $tree && env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tree', round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us/step', [(round(k['avg_launch_us'],1), k['launches']) for k in d['roofline']['kernels']])" )
  done
done
