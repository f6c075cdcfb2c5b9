#!/bin/bash
# on the GPU box: many seeds of the linear / mixed sweeps (the AGC walk's rare routes); one summary line per run
# usage: tools/soak.sh <first seed> <last seed>
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/soak; mkdir -p $out; cd $R
for s in $(seq $1 $2); do
  for d in 4 8 16; do
    timeout -k 5 200 python3 -u tools/fuzz_linear.py 500 $s $d > $out/fl_${s}_$d.txt 2>&1; rc=$?
    echo "fuzz_linear seed $s D=$d rc=$rc $(tail -1 $out/fl_${s}_$d.txt | cut -c1-160)" | tee -a $out/summary.txt
    [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1   # (1: a bad case, keep sweeping; anything else: stop)
  done
  timeout -k 5 200 python3 -u tools/fuzz_mixed.py 300 $s > $out/fm_$s.txt 2>&1; rc=$?
  echo "fuzz_mixed seed $s rc=$rc $(tail -1 $out/fm_$s.txt | cut -c1-160)" | tee -a $out/summary.txt
  [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1
done
