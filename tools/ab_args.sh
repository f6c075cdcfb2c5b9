#!/bin/bash
# usage (on the GPU box): tools/ab_args.sh <outfile> "<env>" "<shape_bench args>" ["<env>" "<args>" ...]
out=$1; shift
: > $out
while [ $# -ge 2 ]; do
  echo "[$1] $(env $1 timeout -k 10 150 python3 tools/shape_bench.py $2 2>/dev/null | grep '^FS=')" >> $out
  shift 2
done
cat $out
