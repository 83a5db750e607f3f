#!/bin/bash
# usage: envsweep.sh "A=1" "QSV_TILES_PER_BLOCK=4" "QSV_TILES_PER_BLOCK=8 QSV_STREAMS=1" ...   (one bench run per argument)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 > gpurun_out/sw_$i.json 2> gpurun_out/sw_$i.err || { echo "$setting failed"; tail -3 gpurun_out/sw_$i.err; continue; }
  python - "$setting" gpurun_out/sw_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2]))
print(sys.argv[1], "->", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"]) for k in d["roofline"]["kernels"]], "window", round(d["roofline"]["pass_window_ms_per_step"]*1e3))
PY
done
