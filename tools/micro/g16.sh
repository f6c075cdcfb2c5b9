timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g16_tests.log 2>&1 || { tail -30 gpurun_out/g16_tests.log; exit 1; }
tail -2 gpurun_out/g16_tests.log
for sh in "8000000 131072 131073 256 64 32" "8000000 131072 131073 256 640 32" "20000000 262144 262145 512 1024 32" "64000000 524288 524289 2048 512 16" "64000000 524288 524289 2048 4096 16"; do
timeout -k 10 100 python3 tools/shape_bench.py $sh 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
done
