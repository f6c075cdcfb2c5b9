#!/bin/bash
# usage (on the GPU box): tools/ab_env.sh <outfile> "<shape_bench args>" [ENV=val ...]: one shape under several env settings
out=$1; shift; args=$1; shift
: > $out
echo "default: $(timeout -k 10 150 python3 tools/shape_bench.py $args 2>/dev/null | grep '^FS=')" >> $out
for e in "$@"; do
  echo "$e: $(env $e timeout -k 10 150 python3 tools/shape_bench.py $args 2>/dev/null | grep '^FS=')" >> $out
done
cat $out
