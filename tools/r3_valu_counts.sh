#!/bin/bash
# usage (on the GPU box): tools/r3_valu_counts.sh <tag>: SQ counters per launch of every kernel of bench.py's `shapes`
# workloads -> gpurun_out/<tag>_valu_counts.json (copied to profiles/r03_valu_counts.json: bench.py's roofline_valu reads it)
tag=$1
R=$GRAFT_REPO_ROOT
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES"
for key in c4share c4all c3share c5 c2; do
  out=$R/gpurun_out/${tag}_sq_$key
  mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $out -o pmc -- python3 $R/tools/plan_bench.py $key 6 > $out/pmc.log 2>&1) || exit 1
done
python3 - $R/gpurun_out $tag <<'PY'
import csv, sys, glob, json, collections, os
root, tag = sys.argv[1], sys.argv[2]
res = {}
for key in ("c4share", "c4all", "c3share", "c5", "c2"):
    f = glob.glob(f"{root}/{tag}_sq_{key}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    kern = {}
    for k, d in acc.items():
        if not any(s in k for s in ("kh_chan2", "kh_fm_demod2", "kh_back_a", "kh_fm_finish", "kh_demod_linear", "kh_demod_am", "kh_filter_out")):
            continue
        if cnt[(k, "SQ_WAVES")] < 4:
            continue
        kern[k] = {c: round(v / cnt[(k, c)], 1) for c, v in d.items()}
        kern[k]["launches_averaged"] = cnt[(k, "SQ_WAVES")]
    res[key] = {"source": f"profiles/r03_valu_counts.json: rocprofv3 --pmc (one pass, counters only) -- python3 tools/plan_bench.py {key} 6", "kernels": kern}
json.dump(res, open(f"{root}/{tag}_valu_counts.json", "w"), indent=1)
for key, v in res.items():
    for k, d in v["kernels"].items():
        print(key, k, int(d["SQ_INSTS_VALU"]), int(d["SQ_WAVES"]))
PY
