#!/bin/bash
# usage (on the GPU box): tools/big2.sh <outfile>: the two large wideband shapes only
out=$1; : > $out
run() { timeout -k 10 150 python3 tools/shape_bench.py "$@" 2>/dev/null | grep "^FS=" >> $out || echo "FAILED: $*" >> $out; }
run 20000000 262144 262145 512 256 32       # config 3, one GPU's share of 4
run 64000000 524288 524289 2048 512 16      # config 4, one GPU's share of 8
cat $out
