#!/usr/bin/env python3
"""one of bench.py's `shapes` jobs alone (for rocprofv3): python3 tools/shapes_one.py 0|1|2 [steps]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import torch, pyka9q, bench, workloads
which = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.cuda.set_device(0)
assert pyka9q.lib().ka9q_hip_init(0) == 0
jobs = [
    (workloads.CONFIG4, workloads.shard_plan(workloads.config4_plan(), workloads.CONFIG4, 8, 3)[0], 16, "config 4 share"),
    (workloads.CONFIG3, workloads.shard_plan(workloads.config3_plan(), workloads.CONFIG3, 4, 1)[0], 32, "config 3 share"),
    (workloads.CONFIG5, workloads.config5_plan(), 32, "config 5"),
]
shape, plan, nb, label = jobs[which]
nb = int(os.environ.get("KA9Q_SHAPE_T", nb))  # experiment: blocks per launch
if len(sys.argv) > 3 and sys.argv[3] == "noagc":  # experiment: the linear channels without AGC (no serial walk)
    plan = [dict(ch, agc=0) if ch["demod"] == "linear" else ch for ch in plan]
print(json.dumps(bench.measure_shape(pyka9q, torch, torch.device("cuda:0"), shape, plan, nb, label, steps=steps)))
