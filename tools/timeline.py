#!/usr/bin/env python3
"""print the kernel timeline (start, duration, overlap with the previous kernel) of the last steps of a rocprofv3 kernel trace"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "kh" in r["Kernel_Name"]][-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void kh::", "").replace("kh::", "")[:28]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3s}  {name}")
    prev_end = max(prev_end, e)
