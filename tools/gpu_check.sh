#!/bin/bash
# usage (on the GPU box): tools/gpu_check.sh <tag> [env assignments for extra A/B bench lines ...]
# GPU tests, then bench.py (default env) and one more bench per extra env assignment; prints one summary line each.
tag=$1; shift
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/t_$tag.log
[ $rc -ne 0 ] && exit $rc
summ() { python3 - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1].split('/')[-1], "ms/step", d["ms_per_step"], "MS/s", d["value"], {k:round(v["ms"]*1e3,1) for k,v in d["stages"].items()}, d["roofline"]["kernel"], d["roofline"]["frac"])
PY
}
timeout -k 10 200 python bench.py --no-cpu --no-extra > gpurun_out/bench_$tag.log 2>&1 || { tail -5 gpurun_out/bench_$tag.log; exit 1; }
summ gpurun_out/bench_$tag.log
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --no-cpu --no-extra > gpurun_out/bench_${tag}_$i.log 2>&1 || { tail -5 gpurun_out/bench_${tag}_$i.log; exit 1; }
  echo "$e"; summ gpurun_out/bench_${tag}_$i.log
done
