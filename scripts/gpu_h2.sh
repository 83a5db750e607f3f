#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  label=$1; shift
  env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/h_$label.json 2> gpurun_out/h_$label.err || { echo "$label failed"; tail -3 gpurun_out/h_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/h_$label.json"))
print("$label", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"]) for k in d["roofline"]["kernels"]], "window", round(d["roofline"]["pass_window_ms_per_step"]*1e3))
PY
}
run default A=1
run default_again A=1
run plan_28_36 QSV_PUSH_PLAN=28,36
run plan_36_28 QSV_PUSH_PLAN=36,28
run plan_20_44 QSV_PUSH_PLAN=20,44
run tpb1 QSV_TILES_PER_BLOCK=1
run tpb4 QSV_TILES_PER_BLOCK=4
run streams1 QSV_STREAMS=1
run n24 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32
run n24_64 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=64
run n24_tpb1 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_TILES_PER_BLOCK=1
run n24_tpb4 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_TILES_PER_BLOCK=4
run n24_g16 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_GROUP=16
run n24_g4 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_GROUP=4
