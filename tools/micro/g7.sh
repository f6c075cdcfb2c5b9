FUZZ_ONLY=157 timeout -k 5 200 python3 -u tools/fuzz_bank.py 158 312 2>&1 | tail -4 | cut -c1-300
export KA9Q_HIP_LIB=$GRAFT_REPO_ROOT/ka9q-radio_amd/libka9q_hip_dbg.so
for w in 0 2 8; do echo "== 640 ch, FM2_WAVES=$w"; KA9Q_HIP_FM2_WAVES=$w timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 8000000 131072 131073 256 640 32 || exit 1; done
echo "== 64 ch"; timeout -k 10 100 python3 tools/micro/dbg_fm2_times.py 8000000 131072 131073 256 64 32
