"""Per-launch durations of pass_kernel from a rocprofv3 kernel trace directory (grouped by position in a step)."""
import csv, glob, sys
for root in sys.argv[1:]:
    f = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "pass_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    import os
    per = int(os.environ.get("PT_PER", 6))
    tail = d[-per * 4:]
    cols = [sum(tail[i::per]) / len(tail[i::per]) for i in range(per)]
    print(root, "launches", len(d), "mean us by position in step:", [round(x, 1) for x in cols], "sum", round(sum(cols), 1))
