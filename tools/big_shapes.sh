#!/bin/bash
# usage (on the GPU box): tools/big_shapes.sh <outfile>: the wideband shapes whose forward transform is four-step
out=$1; : > $out
run() { timeout -k 10 150 python3 tools/shape_bench.py "$@" 2>/dev/null | grep "^FS=" >> $out || echo "FAILED: $*" >> $out; }
run 8000000 131072 131073 256 64 32         # configs[1]
run 20000000 262144 262145 512 256 32       # config 3, one GPU's share of 4
run 64000000 524288 524289 2048 512 16      # config 4, one GPU's share of 8
run 64000000 524288 524289 2048 4096 16     # config 4, all channels on one GPU
cat $out
