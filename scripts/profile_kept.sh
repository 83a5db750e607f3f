#!/bin/bash
# rocprofv3 of a last-layer search's evaluations on kept states (scripts/prefix_cache_experiment.py n:layers:population):
# kernel trace + stats, then FETCH_SIZE / WRITE_SIZE passes of their own.   usage: scripts/profile_kept.sh <tag> [spec]
set -u
tag=${1:-r04}; spec=${2:-20:8:64}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp QSV_EXP_ONLY=kept QSV_STREAMS=1
out=gpurun_out/kept_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/prefix_cache_experiment.py $spec > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 scripts/prefix_cache_experiment.py $spec > $out/$c.log 2>&1
done
python3 - $out <<'PY' > $out/summary.txt 2>&1
import csv, glob, sys
root = sys.argv[1]
print("== rocprofv3 --kernel-trace --stats -- python3 scripts/prefix_cache_experiment.py (QSV_EXP_ONLY=kept: 64 kept states made once, then only the last-layer search's evaluations from them; one stream) ==")
for path in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        print(f"  {r['Name'][:74]:74s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:10.1f} total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} pct={r['Percentage']}")
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pass_kernel" in r["Kernel_Name"] and ", false>" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.setdefault(c, []).append(float(r["Counter_Value"]))
for c, v in vals.items():
    v.sort()
    print(f"  later-pass launches: {c} KiB per launch: n={len(v)} min={v[0]:.0f} median={v[len(v)//2]:.0f} max={v[-1]:.0f}")
PY
cat $out/summary.txt; grep '^{' $out/trace.log
