#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench args]  ->  gpurun_out/<tag>/kt (kernel trace + stats)
set -e
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt -o kt -- python3 $R/bench.py --no-cpu --no-extra "$@" > $R/$out/kt.log 2>&1
f=$(find $R/$out/kt -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'kh::' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
