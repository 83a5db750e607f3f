#!/bin/bash
# round 2, call A: GPU tests, default bench line, contended 3x3 EVQE
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/a_tests.log 2>&1
rc=$?
tail -5 gpurun_out/a_tests.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
rc2=$?
echo "bench rc=$rc2"; tail -c 600 gpurun_out/a_bench.err
if [ $rc2 -ne 0 ]; then exit $rc2; fi
timeout -k 10 120 python - > gpurun_out/a_contended.log 2>&1 <<'PY'
import sys, json
sys.path.insert(0, "scripts"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import jssp_instances as inst
from config4 import solve
from queasars_amd.job_shop_scheduling import JSSPDomainWallHamiltonianEncoder
enc = JSSPDomainWallHamiltonianEncoder(inst.three_by_three_contended(), makespan_limit=5, **inst.NOTEBOOK_PENALTIES)
for seed in (0, 1, 2):
    for gens in (8, 16):
        print(json.dumps({"gens": gens, **solve(enc, "sampler", seed, gens)}), flush=True)
PY
echo "contended rc=$?"; tail -3 gpurun_out/a_contended.log
exit $rc
