#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/b_tests.log 2>&1
rc=$?
tail -4 gpurun_out/b_tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 200 python scripts/stamps.py run 20 64 4 > gpurun_out/b_stamps20.log 2>&1 && timeout -k 10 200 python scripts/stamps.py run 24 32 4 > gpurun_out/b_stamps24.log 2>&1
cat gpurun_out/b_stamps20.log gpurun_out/b_stamps24.log | grep -v "^W2026\|^E2026"
exit $rc
