#!/bin/bash
# usage (on the GPU box): tools/pmc_any.sh <tag> "<counters>" <script> [args]  -> gpurun_out/<tag>/ (one PMC pass, no tracing)
tag=$1; shift
ctrs=$1; shift
R=$GRAFT_REPO_ROOT
script=$R/$1; shift
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -o pmc -- python3 $script "$@" > $out/pmc.log 2>&1
f=$(find $out -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].replace('void ','').split('(')[0][:44]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k,d in acc.items():
    if 'kh' not in k: continue
    print(k, {c: round(v/cnt[(k,c)],1) for c,v in d.items()})
PY
