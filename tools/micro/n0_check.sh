# on the GPU box: the compute_n0 tests, then rendezvous against bracket over the four-step shapes
timeout -k 10 600 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_bank_fm.py tests/test_gpu_filter.py tests/test_gpu_stream.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
KA9Q_HIP_N0_DIRECT=2 timeout -k 10 600 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_bank_fm.py tests/test_gpu_zsweeps.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for m in 1 2 1 2; do
echo "== KA9Q_HIP_N0_DIRECT=$m"
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 100 python3 tools/shape_bench.py 8000000 131072 131073 256 64 32 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 100 python3 tools/shape_bench.py 20000000 262144 262145 512 256 32 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 100 python3 tools/shape_bench.py 64000000 524288 524289 2048 512 16 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
KA9Q_HIP_N0_DIRECT=$m timeout -k 10 200 python3 bench.py --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['ms_per_step'], d['value'], {k:(v['ms'],v.get('ms_pipelined')) for k,v in d['stages'].items()})" || exit 1
done
