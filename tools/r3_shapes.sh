#!/bin/bash
# usage (on the GPU box): tools/r3_shapes.sh <tag>: big shapes + kernel stats of config 4 (4096 and 512 channels)
tag=$1
R=$GRAFT_REPO_ROOT
bash tools/big_shapes.sh gpurun_out/${tag}_shapes.txt || exit 1
for c in 4096 512; do
  out=$R/gpurun_out/${tag}_kt_$c
  mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $R/tools/shape_bench.py 64000000 524288 524289 2048 $c 16 > $out/kt.log 2>&1) || exit 1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'kh::' in r['Name'] and int(r['Calls'])>5:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
PY
done
