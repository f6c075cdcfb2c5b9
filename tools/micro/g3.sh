timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g3_tests.log 2>&1 || { tail -30 gpurun_out/g3_tests.log; exit 1; }
tail -2 gpurun_out/g3_tests.log
KA9Q_HIP_N0_DIRECT=2 timeout -k 10 600 python -u -m pytest tests/test_gpu_bank_fm.py tests/test_gpu_configs.py tests/test_gpu_stream.py tests/test_gpu_graph.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g3_tests2.log 2>&1 || { tail -30 gpurun_out/g3_tests2.log; exit 1; }
tail -2 gpurun_out/g3_tests2.log
bash tools/all_shapes.sh gpurun_out/r3_s6_shapes.txt
