#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/g
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/g_tests.log 2>&1
rc=$?
tail -3 gpurun_out/g_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/g_bench.json"))
print(round(d["value"]), "evals/s", [round(k["avg_launch_us"],1) for k in d["roofline"]["kernels"]])
PY
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/g/base -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/g/base.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list))
for path in glob.glob("gpurun_out/g/base/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "pass_kernel" in r["Kernel_Name"]:
            kind = "p0" if ", true>" in r["Kernel_Name"] else "p1"
            acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kind in ("p0","p1"):
    c=acc[kind]; waves=sum(c["SQ_WAVES"])/len(c["SQ_WAVES"])
    print(kind, "waves", round(waves), " per wave:", {k.replace("SQ_",""): round(sum(x)/len(x)/waves,1) for k,x in sorted(c.items()) if k!="SQ_WAVES"})
PY
