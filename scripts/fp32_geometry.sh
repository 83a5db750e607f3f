#!/bin/bash
# fp32 single-gate sweeps under the geometries the pass kernel has (round 4)
for n in 28 26 24; do
  for geo in "0 0" "13 4" "12 4" "12 3" "11 3" "12 2"; do
    set -- $geo
    echo "== n=$n fp32 tile=$1 reg=$2"
    timeout -k 10 120 python scripts/microbench.py --n $n --dtype fp32 --tile-bits $1 --reg-bits $2 --reps 20 --targets 0,3,7,11,12,17,$((n-3)),$((n-1)) 2>/dev/null | tail -1
  done
done
for n in 28 26; do
  echo "== n=$n fp64 default"
  timeout -k 10 120 python scripts/microbench.py --n $n --dtype fp64 --reps 20 --targets 0,3,7,11,12,17,$((n-3)),$((n-1)) 2>/dev/null | tail -1
done
