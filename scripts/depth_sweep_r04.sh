#!/bin/bash
# Throughput by circuit depth (profiles/r04_split_by_depth.txt): bench.py --no-extras with QSV_BENCH_LAYERS=L, one box for the table
cd "$(dirname "$0")/.."
for L in 3 4 5 6 7 8; do
  QSV_BENCH_LAYERS=$L timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        ks = [(round(k['avg_launch_us'], 1), k['launches']) for k in d['roofline']['kernels']]
        print('  L = $L  %8.0f evals/s (resident inputs)  %8.0f (host lists)  %8.0f (fresh inputs)   %s' % (d['value'], d['value_host_lists'] or 0, d['value_fresh_inputs'] or 0, ks))
"
done
QSV_BENCH_QUBITS=24 QSV_BENCH_POP=256 QSV_BENCH_LAYERS=4 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('  n = 24, P = 256, L = 4: %8.0f evals/s (resident inputs)  %8.0f (host lists)' % (d['value'], d['value_host_lists'] or 0))
"
