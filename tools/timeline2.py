#!/usr/bin/env python3
"""kernel timeline of the timed region of a bench.py rocprofv3 trace: steps [skip, skip+n) counted from the end"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "kh" in r["Kernel_Name"]]
tail, n = int(sys.argv[2]), int(sys.argv[3])
sel = rows[-(tail + n):-tail]
t0 = int(sel[0]["Start_Timestamp"]); prev = t0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void kh::", "").replace("kh::", "")[:26]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:6.1f} gap {(s - prev) / 1e3:7.1f} q{r['Queue_Id']} {name}")
    prev = max(prev, e)
