#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/f
for v in base nogates noswap nothing_but_memory; do
  if [ $v != base ]; then export QSV_LIBRARY=$PWD/queasars_amd/libqsv_abl_$v.so; else unset QSV_LIBRARY; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/f/$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/f/$v.log 2>&1 || { echo "$v failed"; tail -5 gpurun_out/f/$v.log; exit 1; }
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for v in ("base","nogates","noswap","nothing_but_memory"):
    acc=defaultdict(lambda: defaultdict(list))
    for path in glob.glob(f"gpurun_out/f/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pass_kernel" in r["Kernel_Name"]:
                kind = "p0" if ", true>" in r["Kernel_Name"] else "p1"
                acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kind in ("p0","p1"):
        c=acc[kind]
        waves=sum(c["SQ_WAVES"])/len(c["SQ_WAVES"])
        print(v, kind, "waves", round(waves), " per wave:", {k.replace("SQ_",""): round(sum(x)/len(x)/waves,1) for k,x in sorted(c.items()) if k!="SQ_WAVES"})
PY
