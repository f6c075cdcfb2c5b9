#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/...) into the small summaries kept under profiles/.

    python tools/summarize_prof.py <kernel-trace dir> <FETCH_SIZE pmc dir> <WRITE_SIZE pmc dir> <out prefix>

Writes <prefix>_kernel_stats.csv (the --stats table restricted to this library's kernels) and <prefix>_hbm_traffic.json
(HBM bytes per launch per kernel from the TCC counters, corrected as /opt/skills/guides/MI355X_MICROARCH.md "HBM"
prescribes for gfx950: FETCH_SIZE is in KiB and reports half of a wide coalesced read stream -> x2; WRITE_SIZE is
exact for 16-byte-per-lane stores)."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def pmc(d, counter):
    out = collections.defaultdict(list)
    f = find(d, "_counter_collection.csv") if d else None
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if "kh::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    kt, fd, wd, prefix = sys.argv[1:5]
    stats = find(kt, "_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(stats)) if "kh::" in r["Name"]]
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    fetch, write = pmc(fd, "FETCH_SIZE"), pmc(wd, "WRITE_SIZE")
    avg = {short(r["Name"]): float(r["AverageNs"]) for r in rows}
    out = {}
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0)  # KiB -> bytes, x2 gfx950 correction
        wr = 1024.0 * write.get(k, 0.0)
        out[k] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr), "avg_launch_ns_kernel_trace": avg.get(k)}
    json.dump({"note": "FETCH_SIZE KiB x2 (gfx950 half-count of wide reads), WRITE_SIZE KiB x1; separate --pmc passes",
               "kernels": out}, open(prefix + "_hbm_traffic.json", "w"), indent=1)
    print(open(prefix + "_hbm_traffic.json").read())


if __name__ == "__main__":
    main()
