#!/bin/bash
# rocprofv3 of the deep rows of bench.py, one row per process (`bench.py --only <row>`: the row's launches on one stream):
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in PMC passes of their own (gpurun wants them separate).
# usage: scripts/profile_rows.sh <tag> [row ..]      -> gpurun_out/rows_<tag>/{summary.txt, traffic.json}
set -u
tag=${1:-r03}; shift || true
rows=${*:-deep_n20_L8 deep_n24_L4_nosplit deep_n24_L8}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/rows_$tag
rm -rf $out; mkdir -p $out
for row in $rows; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${row}_trace -- python3 bench.py --only $row > $out/${row}_trace.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${row}_$c -- python3 bench.py --only $row > $out/${row}_$c.log 2>&1
  done
done
python3 scripts/summarize_rows.py $out $rows > $out/summary.txt 2>&1
cat $out/summary.txt
