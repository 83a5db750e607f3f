#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/q_tests.log 2>&1
rc=$?; tail -3 gpurun_out/q_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() {
  label=$1; shift
  env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/q_$label.json 2> gpurun_out/q_$label.err || { echo "$label failed"; tail -3 gpurun_out/q_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/q_$label.json"))
print("$label", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"]) for k in d["roofline"]["kernels"]], "window", round(d["roofline"]["pass_window_ms_per_step"]*1e3))
PY
}
run default A=1
run default2 A=1
run plan64 QSV_PUSH_PLAN=64
run streams1 QSV_STREAMS=1
run n24 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32
run n16 QSV_BENCH_QUBITS=16 QSV_BENCH_POP=64
run n12 QSV_BENCH_QUBITS=12 QSV_BENCH_POP=64
