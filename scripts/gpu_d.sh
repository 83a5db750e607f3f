#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "geometries or compact" > gpurun_out/d_tests.log 2>&1
rc=$?
tail -3 gpurun_out/d_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() {
  # $1 label, rest: env assignments
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/d_$label.json 2> gpurun_out/d_$label.err || { echo "$label failed"; tail -3 gpurun_out/d_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/d_$label.json"))
print("$label", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"]) for k in d["roofline"]["kernels"]])
PY
}
run n20_k12r3 QSV_TILE_BITS=12 QSV_REG_BITS=3 || exit 1
run n20_k12r4 QSV_TILE_BITS=12 QSV_REG_BITS=4 || exit 1
run n20_k13r4 QSV_TILE_BITS=13 QSV_REG_BITS=4 || exit 1
run n20_k11r3 QSV_TILE_BITS=11 QSV_REG_BITS=3 || exit 1
run n20_k11r4 QSV_TILE_BITS=11 QSV_REG_BITS=4 || exit 1
run n24_k12r3 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_TILE_BITS=12 QSV_REG_BITS=3 || exit 1
run n24_k12r4 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_TILE_BITS=12 QSV_REG_BITS=4 || exit 1
run n24_k13r4 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_TILE_BITS=13 QSV_REG_BITS=4 || exit 1
