#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  label=$1; shift
  env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/j_$label.json 2> gpurun_out/j_$label.err || { echo "$label failed"; tail -3 gpurun_out/j_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/j_$label.json"))
print("$label", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"]) for k in d["roofline"]["kernels"]])
PY
}
for n in 18 21 22 23 25 26 28; do
  pop=32; [ $n -le 21 ] && pop=64; [ $n -ge 26 ] && pop=8; [ $n -ge 28 ] && pop=4
  run n${n}_k12r3 QSV_BENCH_QUBITS=$n QSV_BENCH_POP=$pop
  run n${n}_k13r4 QSV_BENCH_QUBITS=$n QSV_BENCH_POP=$pop QSV_TILE_BITS=13 QSV_REG_BITS=4
  run n${n}_k12r4 QSV_BENCH_QUBITS=$n QSV_BENCH_POP=$pop QSV_TILE_BITS=12 QSV_REG_BITS=4
done
