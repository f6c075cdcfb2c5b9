#!/bin/bash
# usage (on the GPU box): [ENV=..] tools/r3_kt.sh <tag> <shape_bench args...>: kernel stats of one shape
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $R/tools/shape_bench.py "$@" > $out/kt.log 2>&1) || exit 1
grep "^FS=" $out/kt.log
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'kh::' in r['Name'] and int(r['Calls'])>5 and int(r['Calls'])<100:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
