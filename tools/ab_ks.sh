#!/bin/bash
# usage (on the GPU box): tools/ab_ks.sh <tag> [runs]  -- bench.py's headline with the whole spectrum kept and with kh_bank_keep_spectrum(bank, 0), interleaved
tag=$1; runs=${2:-2}
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $runs); do
  for ks in 1 0; do
    KA9Q_HIP_KEEP_SPECTRUM=$ks timeout -k 10 200 python bench.py --no-cpu --no-extra > gpurun_out/bench_${tag}_ks${ks}_$r.log 2>&1 || { tail -5 gpurun_out/bench_${tag}_ks${ks}_$r.log; exit 1; }
    python3 - gpurun_out/bench_${tag}_ks${ks}_$r.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1].split('/')[-1], "us/step", round(d["ms_per_step"]*1e3,2), "long", round(d.get("ms_per_step_long",0)*1e3,2), {k:(round(v["ms"]*1e3,1), round(v.get("ms_pipelined",0)*1e3,1)) for k,v in d["stages"].items()})
PY
  done
done
