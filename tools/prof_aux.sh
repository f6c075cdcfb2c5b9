#!/bin/bash
# usage (on the GPU box): tools/prof_aux.sh <tag> -> gpurun_out/<tag>/: rocprofv3 kernel stats of the packet (AFSK), tone and
# front-end kernels under their throughput tools
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/afsk -o kt -- python3 $R/tools/afsk_bench.py 1024 32 > $out/afsk.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tones -o kt -- python3 $R/tools/tones_bench.py 4096 20 > $out/tones.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fe -o kt -- python3 $R/tools/frontend_bench.py > $out/fe.log 2>&1 || exit 1
tail -2 $out/afsk.log $out/tones.log $out/fe.log
for d in afsk tones fe; do echo "== $d"; cut -d, -f1-4 $out/$d/kt_kernel_stats.csv | head -8; done
