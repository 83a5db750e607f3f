#!/bin/bash
# round-3 starting point: the deep (unsplit) multi-pass path at n = 20 and n = 24, a few geometries
cd "$(dirname "$0")/.."
STEPS=20 bash scripts/envsweep.sh \
  "QSV_BENCH_LAYERS=8" \
  "QSV_BENCH_LAYERS=8 QSV_TILE_BITS=13" \
  "QSV_BENCH_LAYERS=8 QSV_STREAMS=1" \
  "QSV_BENCH_LAYERS=6 QSV_SPLIT=0" \
  "QSV_BENCH_LAYERS=6 QSV_SPLIT=0 QSV_TILE_BITS=13" \
  "QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_SPLIT=0" \
  "QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_SPLIT=0 QSV_STREAMS=1" \
  "QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_BENCH_LAYERS=8" \
  "QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_BENCH_LAYERS=8 QSV_STREAMS=1" \
  "A=1" 2>&1 | tee gpurun_out/r03_baseline.txt
