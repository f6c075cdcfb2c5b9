#!/usr/bin/env python3
"""how long the host takes to enqueue K steps (no sync) vs how long the device takes to finish them"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import bench, torch, pyka9q
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
assert pyka9q.lib().ka9q_hip_init(0) == 0
import workloads
plan = workloads.config2_plan(64)
x = bench.make_wideband_torch(bench.SH, plan, bench.T * bench.L, dev); torch.cuda.synchronize()
T, L = bench.T, bench.L
bank = bench.build_bank(pyka9q, bench.SH, plan, T)
bank.keep_audio(0); bank.keep_baseband(0)
bank.write_input_device(0, x.data_ptr(), T * L)
def step():
    bank.commit_input(0, T * L); bank.execute(T, sync=False)
for _ in range(5): step()
bank.sync()
import ctypes
lib = pyka9q.lib()
K = 3
h = []
for rep in range(6):  # short bursts: never throttled by the pinned-slot ring
    bank.sync()
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    h.append(1e6 * (t1 - t0) / K)
print("host-only enqueue per step (bursts of 3):", [round(v, 1) for v in h])
K = 40
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
bank.sync()
t2 = time.perf_counter()
print(f"env {os.environ.get('KA9Q_HIP_PIPELINE','-')}/{os.environ.get('KA9Q_HIP_GRAPH','-')}: host enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
bank.close()
