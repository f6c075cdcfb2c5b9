#!/bin/bash
# on the GPU box: the bank sweeps with banks that keep only their channels' rows of the spectrum (KA9Q_HIP_KEEP_SPECTRUM=0:
# kh_bank_keep_spectrum's default), with and without the fall-back to whole spectra.  usage: tools/soak_sparse.sh <first seed> <budget seconds>
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/soak_sparse; mkdir -p $out; cd $R
t0=$(date +%s); s=$1
export KA9Q_HIP_KEEP_SPECTRUM=0
while [ $(( $(date +%s) - t0 )) -lt $2 ]; do
  for hold in 64 0; do
    KA9Q_HIP_SPARSE_HOLD=$hold timeout -k 5 280 python3 -u tools/fuzz_bank.py 300 $s > $out/fb_${s}_h$hold.txt 2>&1; rc=$?
    echo "fuzz_bank seed $s keep_spectrum=0 hold=$hold rc=$rc $(tail -1 $out/fb_${s}_h$hold.txt | cut -c1-200)" >> $out/summary.txt
    [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1   # (1: a bad case, keep sweeping; anything else: stop)
  done
  timeout -k 5 200 python3 -u tools/fuzz_mixed.py 300 $s > $out/fm_$s.txt 2>&1; rc=$?
  echo "fuzz_mixed seed $s keep_spectrum=0 rc=$rc $(tail -1 $out/fm_$s.txt | cut -c1-160)" >> $out/summary.txt
  [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1
  s=$((s + 1))
  echo "seed $((s - 1)) done"
done
grep -c "rc=0" $out/summary.txt; grep -v "rc=0" $out/summary.txt | cut -c1-300
