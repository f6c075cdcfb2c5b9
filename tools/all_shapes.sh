#!/bin/bash
# usage (on the GPU box): tools/all_shapes.sh <outfile>: tools/shape_bench.py over the BASELINE shapes (FM on every channel unless noted)
out=$1; : > $out
run() { timeout -k 10 150 python3 tools/shape_bench.py "$@" 2>/dev/null | grep "^FS=" >> $out || echo "FAILED: $*" >> $out; }
run 192000 3840 4353 4 1 32                 # config 1
run 192000 3840 3841 4 1 32                 # reference default sizes (N = 7680)
run 8000000 131072 131073 256 64 32         # configs[1]
run 8000000 131072 131073 256 640 32        # the whole 8 MS/s raster
run 20000000 262144 262145 512 256 32       # config 3, one GPU's share of 4
run 20000000 262144 262145 512 1024 32      # config 3, all channels on one GPU
run 64000000 524288 524289 2048 512 16      # config 4, one GPU's share of 8
run 64000000 524288 524289 2048 4096 16     # config 4, all channels on one GPU
run 192000 3840 4353 4 256 32 256           # config 5 shape with FM
run 192000 3840 4353 4 256 32 256 linear    # config 5: linear + PLL + AGC
cat $out
