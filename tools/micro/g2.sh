export KA9Q_HIP_LIB=$GRAFT_REPO_ROOT/ka9q-radio_amd/libka9q_hip_dbg.so
timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 16000000 131072 131073 512 64 32 && timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 64000000 524288 524289 2048 512 16 && timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 64000000 524288 524289 2048 4096 16
