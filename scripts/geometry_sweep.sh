#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  label=$1; shift
  env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/geo_$label.json 2> gpurun_out/geo_$label.err || { echo "$label failed"; tail -3 gpurun_out/geo_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/geo_$label.json"))
print("$label", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"]) for k in d["roofline"]["kernels"]])
PY
}
for n in 14 16 19 20 24; do
  pop=64; [ $n -ge 24 ] && pop=32
  run n${n}_k12r3 QSV_BENCH_QUBITS=$n QSV_BENCH_POP=$pop
  run n${n}_k12r4 QSV_BENCH_QUBITS=$n QSV_BENCH_POP=$pop QSV_TILE_BITS=12 QSV_REG_BITS=4
  run n${n}_k13r4 QSV_BENCH_QUBITS=$n QSV_BENCH_POP=$pop QSV_TILE_BITS=13 QSV_REG_BITS=4
done
run n20_k12r3_again QSV_BENCH_QUBITS=20
