#!/bin/bash
# on the GPU box: the linear demodulator's tests and sweeps, then config 3's share in isolation and pipelined (kernel trace)
R=$GRAFT_REPO_ROOT; tag=${1:-lw}; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_bank_linear.py tests/test_gpu_configs.py tests/test_gpu_zsweeps.py -x -q --timeout 180 --timeout-method thread > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -1 $out/tests.txt
for s in 7 31 41; do timeout -k 5 200 python3 -u tools/fuzz_linear.py 400 $s > $out/fl_$s.txt 2>&1; echo "fuzz_linear $s rc=$? $(tail -1 $out/fl_$s.txt)"; done
timeout -k 5 200 python3 -u tools/fuzz_mixed.py 300 7 > $out/fm.txt 2>&1; echo "fuzz_mixed rc=$? $(tail -1 $out/fm.txt)"
cd /tmp && export TMPDIR=/tmp
KA9Q_HIP_PIPELINE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/iso -o kt -- python3 $R/tools/shapes_one.py 1 > $out/iso.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/pipe -o kt -- python3 $R/tools/shapes_one.py 1 > $out/pipe.txt 2>&1 || exit 1
cd $R && python3 - <<PY
import csv, re
for w in ("iso", "pipe"):
    rows = list(csv.DictReader(open("$out/%s/kt_kernel_trace.csv" % w)))
    d = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "linear_w" in r["Kernel_Name"]]
    print(w, "linear_w us:", d)
    print(w, re.findall(r'"us_per_step": [0-9.]+|"stage_us_hip_events": {[^}]*}', open("$out/%s.txt" % w).read()))
PY
