"""Kernel timeline of the last step in a rocprofv3 kernel trace: start / end (us) per kernel with its queue.  A step is what
lies between two idle gaps of the GPU longer than `gap_us` (default 12): usage timeline.py <trace dir> [gap_us]"""
import csv, glob, os, sys
root = sys.argv[1]
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
f = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [0]
busy_until = int(rows[0]["End_Timestamp"])
for i, r in enumerate(rows[1:], 1):
    if int(r["Start_Timestamp"]) - busy_until > gap_us * 1e3:
        starts.append(i)
    busy_until = max(busy_until, int(r["End_Timestamp"]))
lo, hi = (starts[-2], starts[-1]) if len(starts) > 2 else (starts[-1], len(rows))  # (the last complete step)
if len(sys.argv) > 3:  # (steps that leave no idle gap: just the last so many launches)
    lo, hi = max(0, len(rows) - int(sys.argv[3])), len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
if lo:
    print(f"gap since the end of the step before: {(t0 - max(int(r['End_Timestamp']) for r in rows[:lo])) / 1e3:.1f} us")
for r in rows[lo:hi]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  q{r['Queue_Id']:>2s}  {name}  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r.get('Grid_Size_Z', '1')}")
