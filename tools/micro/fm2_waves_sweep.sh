for w in 0 4 8 16; do
echo "== FM2_WAVES=$w"
for sh in "8000000 131072 131073 256 64 32" "8000000 131072 131073 256 640 32" "20000000 262144 262145 512 256 32" "20000000 262144 262145 512 1024 32" "64000000 524288 524289 2048 512 16" "64000000 524288 524289 2048 4096 16"; do
KA9Q_HIP_FM2_WAVES=$w timeout -k 10 100 python3 tools/shape_bench.py $sh 2>/dev/null | grep "^FS=" | sed 's/FS=[0-9]* L=[0-9]* M=[0-9]* //' || exit 1
done
done
