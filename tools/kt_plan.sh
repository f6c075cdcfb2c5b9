#!/bin/bash
# usage (on the GPU box): tools/kt_plan.sh <tag> <plan key> [env assignments...]: kernel trace of tools/plan_bench.py <key>, one line per kernel
tag=$1; key=$2; shift 2
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/kt_$tag
mkdir -p $out
for e in "$@"; do export "$e"; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $R/tools/plan_bench.py $key 200 > $out/kt.log 2>&1) || { tail -5 $out/kt.log; exit 1; }
grep "^PLAN=" $out/kt.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'kh::' in r['Name'] and int(r['Calls'])>5:
        print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
find $out -name '*kernel_trace.csv' -delete
