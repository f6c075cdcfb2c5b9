timeout -k 10 600 python -u -m pytest tests/test_gpu_bank_fm.py tests/test_gpu_configs.py tests/test_gpu_c_abi.py tests/test_gpu_zsweeps.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g1_tests.log 2>&1 || { tail -30 gpurun_out/g1_tests.log; exit 1; }
tail -2 gpurun_out/g1_tests.log
for c in 0; do echo cpw $c; KA9Q_HIP_LIB=$GRAFT_REPO_ROOT/ka9q-radio_amd/libka9q_hip_dbg.so KA9Q_HIP_FM2_CPW=$c timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 4096 16 || exit 1; done
for c in 1 0; do KA9Q_HIP_FM2_CPW=$c timeout -k 10 100 python3 tools/shape_bench.py 64000000 524288 524289 2048 4096 16 || exit 1; done
timeout -k 10 100 python3 tools/shape_bench.py 64000000 524288 524289 2048 512 16 || exit 1
timeout -k 10 100 python3 tools/shape_bench.py 16000000 131072 131073 512 64 32 || exit 1
