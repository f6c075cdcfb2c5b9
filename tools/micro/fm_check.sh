# on the GPU box: FM bank tests, then the demodulator stage over the bench shapes
timeout -k 10 600 python -u -m pytest tests/test_gpu_bank_fm.py tests/test_gpu_configs.py tests/test_gpu_graph.py tests/test_gpu_zsweeps.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for sh in "8000000 131072 131073 256 64 32" "8000000 131072 131073 256 640 32" "20000000 262144 262145 512 1024 32" "64000000 524288 524289 2048 4096 16"; do
timeout -k 10 100 python3 tools/shape_bench.py $sh 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
done
for k in c2 c3share; do PLAN_STAGES=1 timeout -k 10 100 python3 tools/plan_bench.py $k 20 2>/dev/null | grep -E "PLAN|stages"; done
