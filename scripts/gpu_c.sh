#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/c_tests.log 2>&1
rc=$?
tail -4 gpurun_out/c_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for sw in 1 0; do
  QSV_SWAPS=$sw timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > gpurun_out/c_bench_sw$sw.json 2> gpurun_out/c_bench_sw$sw.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/c_bench_sw$sw.json"))
print("swaps=$sw", round(d["value"]), "evals/s", [round(k["avg_launch_us"],1) for k in d["roofline"]["kernels"]])
PY
done
timeout -k 10 200 python scripts/stamps.py run 20 64 4 > gpurun_out/c_stamps20.log 2>&1 && timeout -k 10 200 python scripts/stamps.py run 24 32 4 > gpurun_out/c_stamps24.log 2>&1
cat gpurun_out/c_stamps20.log gpurun_out/c_stamps24.log | grep -v "^W2026\|^E2026\|amdgpu.ids"
for sw in 1 0; do
QSV_SWAPS=$sw QSV_BENCH_QUBITS=24 QSV_BENCH_POP=32 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/c_bench24_sw$sw.json 2> gpurun_out/c_bench24_sw$sw.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/c_bench24_sw$sw.json"))
print("n=24 swaps=$sw", round(d["value"]), "evals/s", [round(k["avg_launch_us"],1) for k in d["roofline"]["kernels"]])
PY
done
