"""Kernel timeline of the last step in a rocprofv3 kernel trace: start/end (us) per kernel, with stream/queue."""
import csv, glob, sys
root = sys.argv[1]
import os
f = max(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps end with the reduce_partials kernel, or -- split evaluations under a quadratic operator -- with factor_combine_kernel
last = "factor_combine" if any("factor_combine" in r["Kernel_Name"] for r in rows) else "reduce_partials"
ends = [i for i, r in enumerate(rows) if last in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = int(rows[ends[-2]]["End_Timestamp"])
print(f"gap since previous step's last kernel: {(t0 - prev_end) / 1e3:.1f} us")
for r in rows[lo:hi]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  q{r['Queue_Id']:>2s}  {name}  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}")
