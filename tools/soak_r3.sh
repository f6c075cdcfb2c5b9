#!/bin/bash
# on the GPU box: the FM-bank and mixed sweeps (round 3's back half and compute_n0 modes) over seeds until the time budget is used
# usage: tools/soak_r3.sh <first seed> <budget seconds>
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/soak_r3; mkdir -p $out; cd $R
t0=$(date +%s); s=$1
while [ $(( $(date +%s) - t0 )) -lt $2 ]; do
  for mode in -1 2; do
    KA9Q_HIP_N0_DIRECT=$mode timeout -k 5 280 python3 -u tools/fuzz_bank.py 300 $s > $out/fb_${s}_m$mode.txt 2>&1; rc=$?
    echo "fuzz_bank seed $s n0_direct=$mode rc=$rc $(tail -1 $out/fb_${s}_m$mode.txt | cut -c1-200)" | tee -a $out/summary.txt
    [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1   # (1: a bad case, keep sweeping; anything else: stop)
  done
  timeout -k 5 200 python3 -u tools/fuzz_mixed.py 300 $s > $out/fm_$s.txt 2>&1; rc=$?
  echo "fuzz_mixed seed $s rc=$rc $(tail -1 $out/fm_$s.txt | cut -c1-160)" | tee -a $out/summary.txt
  [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1
  s=$((s + 1))
done
