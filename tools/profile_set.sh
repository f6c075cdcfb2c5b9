#!/bin/bash
# usage (on the GPU box): tools/profile_set.sh <tag>  -> gpurun_out/<tag>/{kt,fetch,write} + gpurun_out/<tag>/bench.json
# kernel trace + stats, then the two HBM counter passes (each its own run, counters only), then the plain bench line.
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $R/bench.py --no-cpu --no-extra > $out/kt.log 2>&1 || exit 1
KA9Q_BENCH_PREWARM_S=0 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 $R/bench.py --no-cpu --no-extra --steps 5 --warmup 2 > $out/fetch.log 2>&1 || exit 1
KA9Q_BENCH_PREWARM_S=0 timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 $R/bench.py --no-cpu --no-extra --steps 5 --warmup 2 > $out/write.log 2>&1 || exit 1
cd $R && timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
python3 tools/summarize_prof.py $out/kt $out/fetch $out/write $out/summary > /dev/null && cat $out/bench.json
