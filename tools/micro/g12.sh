for pr in 1 2 0; do
echo "== KA9Q_HIP_PRIO=$pr"
for k in c4all c4share c3share; do KA9Q_HIP_PRIO=$pr timeout -k 10 100 python3 tools/plan_bench.py $k 20 2>/dev/null | grep -E "^PLAN=" | cut -c1-260 || exit 1; done
KA9Q_HIP_PRIO=$pr timeout -k 10 200 python3 bench.py --no-cpu --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['ms_per_step'], d['value'], {k:(v['ms'],v.get('ms_pipelined')) for k,v in d['stages'].items()})" || exit 1
done
