#!/bin/bash
# ISA of the default pass kernel (fp64, R = 3, exchange mode 2) and its instruction mix; no GPU needed.
#   scripts/isa_probe.sh [extra hipcc flags]   ->  /tmp/qsv_probe.s
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQSV_PROBE_ONLY "$@" -S --cuda-device-only \
    queasars_amd/csrc/kernels.hip -o /tmp/qsv_probe_all.s -Rpass-analysis=kernel-resource-usage 2> /tmp/qsv_probe.log || { cat /tmp/qsv_probe.log; exit 1; }
start=$(grep -n '^_ZN3qsv11pass_kernelIdLi3ELi2EE.*:' /tmp/qsv_probe_all.s | head -1 | cut -d: -f1)
awk -v s="$start" 'NR>=s{print} NR>s && /s_endpgm/{exit}' /tmp/qsv_probe_all.s > /tmp/qsv_probe.s
grep -A14 'pass_kernelIdLi3ELi2EE' /tmp/qsv_probe.log | grep -E ' VGPRs:|SGPRs:|Spill|Scratch|Occupancy' | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ';'; echo
for i in v_fma_f64 v_mul_f64 v_mov_b64 v_mov_b32 v_cndmask s_cbranch s_load s_waitcnt v_readlane v_writelane scratch_; do
    printf "%s=%s " $i $(grep -c "$i" /tmp/qsv_probe.s); done; echo
