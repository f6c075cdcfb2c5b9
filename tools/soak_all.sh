#!/bin/bash
# on the GPU box: the other sweeps over a range of seeds; one summary line per run
# usage: tools/soak_all.sh <first seed> <last seed>
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/soak_all; mkdir -p $out; cd $R
for s in $(seq $1 $2); do
  for t in "fuzz_bank 300" "fuzz_filter 300" "fuzz_ingest 300" "fuzz_frontend 400" "fuzz_rtp 300" "fuzz_afsk 40" "fuzz_tones 60"; do
    set -- $t
    timeout -k 5 280 python3 -u tools/$1.py $2 $s > $out/$1_$s.txt 2>&1; rc=$?
    echo "$1 seed $s rc=$rc $(tail -1 $out/$1_$s.txt | cut -c1-200)" | tee -a $out/summary.txt
    [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit 1
  done
done
