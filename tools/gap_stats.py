#!/usr/bin/env python3
"""usage: tools/gap_stats.py <kernel_trace.csv> [skip]: where a step's time goes on each hardware queue of a rocprofv3
--kernel-trace of tools/plan_bench.py -- per queue the busy time, and per kernel the idle gap between the end of the
kernel before it on the same queue and its own start (median / mean), over the steady part of the run (the first `skip`
dispatches of every queue dropped)."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
byq = {}
for r in rows:
    if 'kh::' not in r['Kernel_Name']:
        continue
    byq.setdefault(r['Queue_Id'], []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48]))
for q, v in sorted(byq.items()):
    v.sort()
    v = v[skip:]
    if len(v) < 20:
        continue
    span = v[-1][1] - v[0][0]
    busy = sum(e - s for s, e, _ in v)
    print(f"queue {q}: {len(v)} dispatches over {span/1e3:.0f} us, busy {busy/span:.3f}")
    gaps, durs = {}, {}
    for (s0, e0, n0), (s1, e1, n1) in zip(v, v[1:]):
        gaps.setdefault((n0, n1), []).append(s1 - e0)
        durs.setdefault(n1, []).append(e1 - s1)
    for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(g) < 10:
            continue
        print(f"   {a:48s} -> {b:48s} n {len(g):5d} gap median {statistics.median(g)/1e3:7.2f} mean {statistics.mean(g)/1e3:7.2f} us;"
              f" {b.split('::')[-1][:20]} runs {statistics.mean(durs[b])/1e3:7.2f}")
