#!/bin/bash
# usage (on the GPU box): tools/kt_gaps.sh <tag> <plan key> [env assignments...]: kernel trace of tools/plan_bench.py <key>, then
# tools/gap_stats.py over it (busy fraction of each hardware queue, idle gaps between consecutive kernels)
tag=$1; key=$2; shift 2
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gaps_$tag
mkdir -p $out
for e in "$@"; do export "$e"; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 $R/tools/plan_bench.py $key 300 > $out/kt.log 2>&1) || { tail -5 $out/kt.log; exit 1; }
grep "^PLAN=" $out/kt.log
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 $R/tools/gap_stats.py "$f" 300 > $out/gaps.txt
find $out -name '*kernel_trace.csv' -delete
cat $out/gaps.txt
