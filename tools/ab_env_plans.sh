#!/bin/bash
# usage (on the GPU box): tools/ab_env_plans.sh <outfile> "<plan keys>" <env assignment or ''> ...: tools/plan_bench.py of each key under each environment, interleaved, twice
out=$1; keys=$2; shift 2; : > $out
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for key in $keys; do
    for e in "$@"; do
      echo "-- env: $e" >> $out
      ( for a in $e; do export "$a"; done; PLAN_STAGES=1 timeout -k 10 150 python3 tools/plan_bench.py $key 200 2>/dev/null | grep -E "^PLAN=|^stages" ) >> $out || echo "FAILED $key $e" >> $out
    done
  done
done
cat $out
