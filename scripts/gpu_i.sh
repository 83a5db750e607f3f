#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  label=$1; shift
  env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/i_$label.json 2> gpurun_out/i_$label.err || { echo "$label failed"; tail -3 gpurun_out/i_$label.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/i_$label.json"))
print("$label", round(d["value"]), "evals/s", [(round(k["avg_launch_us"],1), k["launches"], round(k["frac_hbm_algorithmic"],2)) for k in d["roofline"]["kernels"]], "window", round(d["roofline"]["pass_window_ms_per_step"]*1e3))
PY
}
run n24_k12r3 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32
run n24_k13r4 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_TILE_BITS=13 QSV_REG_BITS=4
run n24_k12r3_s1 QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_STREAMS=1
run n24_k12r3_noswap QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 QSV_SWAPS=0
run n16 QSV_BENCH_QUBITS=16 QSV_BENCH_POP=256
run n12 QSV_BENCH_QUBITS=12 QSV_BENCH_POP=256
timeout -k 10 300 python bench.py > gpurun_out/i_bench_full.json 2> gpurun_out/i_bench_full.err; echo full rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/i_bench_full.json"))
print(round(d["value"]), d["config3"]["value"], d["cold_structure_evals_per_s"], d["threaded_b1_evals_per_s"], d["cpu_baseline"]["value"])
PY
