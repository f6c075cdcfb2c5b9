for c in "0 0" "5 0" "10 0" "15 0" "0 5" "0 10" "0 15" "10 10"; do set -- $c
echo "== stagger cols $1 rows $2"
KA9Q_HIP_FFT_STAGGER_C=$1 KA9Q_HIP_FFT_STAGGER_R=$2 timeout -k 10 100 python3 tools/shape_bench.py 8000000 131072 131073 256 64 32 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
done
