timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/g13_tests.log 2>&1 || { tail -30 gpurun_out/g13_tests.log; exit 1; }
tail -2 gpurun_out/g13_tests.log
timeout -k 10 300 python3 bench.py > gpurun_out/g13_bench.json 2> gpurun_out/g13_bench.err || { tail -5 gpurun_out/g13_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/g13_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['value_long'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['frac_alone'], d['roofline']['traffic_source'])
print({k:(v['ms'],v.get('ms_pipelined')) for k,v in d['stages'].items()})
for s in d['shapes']: print(s['channels'], s['N'], s['us_per_step'], s['frac_of_hbm_peak'])
"
bash tools/all_shapes.sh gpurun_out/r3_s8_shapes.txt
