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
run plan_16_24_24 QSV_PUSH_PLAN=16,24,24
run plan_8_24_32 QSV_PUSH_PLAN=8,24,32
run plan_12_20_32 QSV_PUSH_PLAN=12,20,32
run plan_8_16_40 QSV_PUSH_PLAN=8,16,40
run plan_16_48 QSV_PUSH_PLAN=16,48
run plan_8_56 QSV_PUSH_PLAN=8,56
run plan_64 QSV_PUSH_PLAN=64
run plan_32_32 QSV_PUSH_PLAN=32,32
run plan_16x4 QSV_PUSH_PLAN=16,16,16,16
run plan_8_8_16_32 QSV_PUSH_PLAN=8,8,16,32
run plan_4_12_16_32 QSV_PUSH_PLAN=4,12,16,32
run plan_24_40 QSV_PUSH_PLAN=24,40
python scripts/hosttime2.py 2>&1 | grep -v amdgpu.ids
