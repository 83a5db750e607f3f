#!/bin/bash
# usage: deep_ab.sh "QSV_FUSE=0" "QSV_FUSE=1" ...   -- the deep rows of bench.py (one stream, kernels alone) once per setting
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for setting in "$@"; do
  i=$((i+1))
  for row in ${ROWS:-deep_n20_L8 deep_n24_L4_nosplit deep_n24_L8}; do
    env $setting timeout -k 10 200 python bench.py --only $row > gpurun_out/ab_${i}_$row.json 2> gpurun_out/ab_${i}_$row.err || { echo "$setting $row failed"; tail -3 gpurun_out/ab_${i}_$row.err; continue; }
    python - "$setting" $row gpurun_out/ab_${i}_$row.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["deep"][sys.argv[2]]
ks=[(k["kernel"].split("(")[-1][:14], round(k["avg_launch_us"],1), k["launches"]) for k in d["kernels_one_stream"]]
print(f"{sys.argv[1]:28s} {sys.argv[2]:22s} {d['value']:9.0f} evals/s (one stream {d['value_one_stream']:9.0f})  later frac_hbm {d['later_pass_frac_hbm']}", ks, flush=True)
PY
  done
done
