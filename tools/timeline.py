#!/usr/bin/env python3
"""Print the kernel timeline of a rocprofv3 --kernel-trace CSV: per kernel start / duration (us) relative to the first
kernel, its queue, grid -- to see which chain bounds an epoch.   python tools/timeline.py trace.csv [first_row] [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
for r in rows[lo:lo + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    name = name.split("(")[0].replace("gx::", "").replace("void ", "")[:58]
    print(f"{(s - t0) / 1e3:10.1f} +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>2} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4}  {name}")
