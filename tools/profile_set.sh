#!/bin/bash
# usage (on the GPU box): tools/profile_set.sh <tag> [script args...]
#   -> gpurun_out/<tag>/{kt,fetch,write} + gpurun_out/<tag>/summary_{kernel_stats.csv,hbm_traffic.json} (+ bench.json)
# kernel trace + stats, then the two HBM counter passes (each its own run, counters only), then the plain bench line.
# Without script args the profiled command is bench.py's headline run; with them it is `python3 <script args>` (e.g.
# tools/shape_bench.py 64000000 524288 524289 2048 512 16 for config 4's share) and no bench line is taken.
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ $# -gt 0 ]; then
  script=$R/$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $script "$@" > $out/kt.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 $script "$@" > $out/fetch.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 $script "$@" > $out/write.log 2>&1 || exit 1
  cd $R && python3 tools/summarize_prof.py $out/kt $out/fetch $out/write $out/summary > /dev/null && grep -E "^(FS|PLAN)=" $out/kt.log
  rc=$?
  find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -delete  # (tens of MB; the summaries are what is kept)
  exit $rc
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $R/bench.py --no-cpu --no-extra > $out/kt.log 2>&1 || exit 1
KA9Q_BENCH_PREWARM_S=0 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 $R/bench.py --no-cpu --no-extra --steps 5 --warmup 2 > $out/fetch.log 2>&1 || exit 1
KA9Q_BENCH_PREWARM_S=0 timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 $R/bench.py --no-cpu --no-extra --steps 5 --warmup 2 > $out/write.log 2>&1 || exit 1
cd $R && timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
python3 tools/summarize_prof.py $out/kt $out/fetch $out/write $out/summary > /dev/null && cat $out/bench.json
find $out -name '*kernel_trace.csv' -delete; find $out -name '*counter_collection.csv' -delete
