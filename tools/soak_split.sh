#!/bin/bash
# on the GPU box: the bank sweeps with every FM channel's blocks dealt over 2 and 3 workgroups of kh_fm_demod2 (KA9Q_HIP_FM2_SPLIT;
# the sweeps' launches are too short for the split to come about by itself) and with the back halves queued by the bank's own
# thread (KA9Q_HIP_ENQ_THREAD=1).  usage: tools/soak_split.sh <first seed> <budget seconds>
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/soak_split; mkdir -p $out; cd $R
t0=$(date +%s); s=$1
while [ $(( $(date +%s) - t0 )) -lt $2 ]; do
  for split in 2 3; do
    KA9Q_HIP_FM2_SPLIT=$split KA9Q_HIP_ENQ_THREAD=$((split - 2)) timeout -k 5 280 python3 -u tools/fuzz_bank.py 300 $s > $out/fb_${s}_s$split.txt 2>&1; rc=$?
    echo "fuzz_bank seed $s split=$split enq_thread=$((split - 2)) rc=$rc $(tail -1 $out/fb_${s}_s$split.txt | cut -c1-200)" >> $out/summary.txt
    [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1   # (1: a bad case, keep sweeping; anything else: stop)
  done
  KA9Q_HIP_FM2_SPLIT=2 KA9Q_HIP_ENQ_THREAD=1 timeout -k 5 200 python3 -u tools/fuzz_mixed.py 300 $s > $out/fm_$s.txt 2>&1; rc=$?
  echo "fuzz_mixed seed $s split=2 enq_thread=1 rc=$rc $(tail -1 $out/fm_$s.txt | cut -c1-160)" >> $out/summary.txt
  [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1
  s=$((s + 1))
  echo "seed $((s - 1)) done"
done
grep -c "rc=0" $out/summary.txt; grep -v "rc=0" $out/summary.txt | cut -c1-300
