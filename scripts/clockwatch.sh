#!/bin/bash
# Shader clock and power while the bench workload runs (is the fp64 path clock- or power-limited?).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --no-extras --no-cpu-baseline --steps 150000 --warmup 5 > gpurun_out/cw_bench.json 2> gpurun_out/cw_bench.err &
pid=$!
sleep 14   # import + build of the population
for i in 1 2 3 4 5 6 7 8 9 10; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i -E "sclk|mclk|fclk|power|Temperature \(Sensor (edge|junction)" | tr '\n' ';'
  echo
  sleep 1.5
done > gpurun_out/cw_clocks.txt
wait $pid
cat gpurun_out/cw_clocks.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/cw_bench.json")); print(d["value"], d["ms_per_step"])
PY
