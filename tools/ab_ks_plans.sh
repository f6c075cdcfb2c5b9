#!/bin/bash
# usage (on the GPU box): tools/ab_ks_plans.sh <outfile> [steps]: bench.py's shapes with the whole spectrum kept (1) and with only
# the channels' rows (0), interleaved, twice; per-stage times of each kernel alone
out=$1; steps=${2:-200}; : > $out
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for key in c2 c4share c4all c3share; do
    for ks in 1 0; do
      PLAN_KEEP_SPECTRUM=$ks PLAN_STAGES=1 timeout -k 10 150 python3 tools/plan_bench.py $key $steps 2>/dev/null | grep -E "^PLAN=|^stages" >> $out || echo "FAILED $key $ks" >> $out
    done
  done
done
cat $out
