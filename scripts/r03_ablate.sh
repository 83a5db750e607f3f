#!/bin/bash
# where the deep (unsplit) multi-pass path spends its time: ablation builds + stamped build, n = 24 and n = 20, eight layers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== n=24 P=16 L=8 one stream"
QSV_BENCH_LAYERS=8 QSV_STREAMS=1 timeout -k 10 400 python scripts/ablate.py run 24 16
echo "== n=20 P=64 L=8 one stream"
QSV_BENCH_LAYERS=8 QSV_STREAMS=1 timeout -k 10 300 python scripts/ablate.py run 20 64
echo "== stamps n=24 P=16 L=8"
QSV_STREAMS=1 timeout -k 10 200 python scripts/stamps.py run 24 16 8
echo "== stamps n=20 P=64 L=8"
QSV_STREAMS=1 timeout -k 10 200 python scripts/stamps.py run 20 64 8
} 2>&1 | tee gpurun_out/r03_ablate.txt
