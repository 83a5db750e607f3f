#!/bin/bash
# single precision: 32-byte runs (two low lane bits) against 64-byte runs (three), single-gate sweeps and the config-5 sweep's route through the state
for n in 28 26; do
  for env in "QSV_X=0" "QSV_LOW_BITS=3 QSV_LANE_BITS=3" "QSV_LOW_BITS=3 QSV_LANE_BITS=3 QSV_TILE_BITS=12 QSV_REG_BITS=3" "QSV_TILE_BITS=12 QSV_REG_BITS=3"; do
    echo "== n=$n fp32 $env"
    env $env timeout -k 10 120 python scripts/microbench.py --n $n --dtype fp32 --reps 20 --targets 0,3,7,11,12,17,$((n-3)),$((n-1)) 2>/dev/null | tail -1
  done
done
