for m in 0 1 2; do echo "== n0_direct=$m"; KA9Q_HIP_N0_DIRECT=$m timeout -k 5 200 python3 -u tools/fuzz_bank.py 158 312 2>&1 | grep -A8 "BAD" | cut -c1-260; done
echo "== tests"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g6_tests.log 2>&1 || { tail -30 gpurun_out/g6_tests.log; exit 1; }
tail -2 gpurun_out/g6_tests.log
export KA9Q_HIP_LIB=$GRAFT_REPO_ROOT/ka9q-radio_amd/libka9q_hip_dbg.so
timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 8000000 131072 131073 256 64 32 && timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 64000000 524288 524289 2048 4096 16
unset KA9Q_HIP_LIB
bash tools/all_shapes.sh gpurun_out/r3_s7_shapes.txt
