#!/bin/bash
# rocprofv3 runs of bench.py: kernel trace + stats, then PMC passes (kept separate, as gpurun requires).
# usage: scripts/profile.sh <tag> [extra bench args]
set -u
tag=${1:-r02}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp QSV_BENCH_WINDOWS=1 QSV_BENCH_PREWARM_S=0.02
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/bench_trace.log 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out/bench_pmc_$name.log 2>&1
done
python3 scripts/summarize_profile.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
