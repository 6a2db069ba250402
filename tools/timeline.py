"""Prints a steady-state slice of a rocprofv3 kernel trace as a timeline (us).
usage: python tools/timeline.py <dir containing *kernel_trace.csv> [first_row] [rows]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s q%-2s start %9.2f  end %9.2f  dur %7.2f  wgs %s" % (
        r["Kernel_Name"].replace("void ", "")[:30], r["Queue_Id"], (s - t0) / 1e3, (e - t0) / 1e3,
        (e - s) / 1e3, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
