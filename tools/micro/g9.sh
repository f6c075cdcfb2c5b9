timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g9_tests.log 2>&1 || { tail -30 gpurun_out/g9_tests.log; exit 1; }
tail -2 gpurun_out/g9_tests.log
for m in -1 2; do
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 100 python3 tools/shape_bench.py 8000000 131072 131073 256 64 32 2>/dev/null | grep "^FS=" || exit 1
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 100 python3 tools/shape_bench.py 20000000 262144 262145 512 256 32 2>/dev/null | grep "^FS=" || exit 1
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 200 python3 bench.py --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['ms_per_step'], d['value'], {k:(v['ms'],v.get('ms_pipelined')) for k,v in d['stages'].items()})" || exit 1
done
