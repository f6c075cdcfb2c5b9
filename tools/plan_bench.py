#!/usr/bin/env python3
"""one of bench.py's `shapes` workloads on its own (the same plan, input and bank as bench.shapes_leg builds), for the
profilers:  python3 tools/plan_bench.py c4share|c3share|c5|c4all|c2 [steps]
   rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/plan_bench.py c4all 6   (tools/r3_valu_counts.sh)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import torch, pyka9q, workloads, bench
key = sys.argv[1]
bench.KEEP_SPECTRUM = int(os.environ.get("PLAN_KEEP_SPECTRUM", "0"))  # 1: the whole spectrum of every block is stored (A/B)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
jobs = {
    "c4share": (workloads.CONFIG4, lambda: workloads.shard_plan(workloads.config4_plan(), workloads.CONFIG4, 8, 3)[0], 16),
    "c3share": (workloads.CONFIG3, lambda: workloads.shard_plan(workloads.config3_plan(), workloads.CONFIG3, 4, 1)[0], 32),
    "c5": (workloads.CONFIG5, workloads.config5_plan, 32),
    "c4all": (workloads.CONFIG4, workloads.config4_plan, 16),
    "c2": (workloads.CONFIG2, lambda: workloads.config2_plan(64), 32),
}
shape, mkplan, nb = jobs[key]
nb = int(os.environ.get("PLAN_BLOCKS", nb))  # blocks per step (A/B)
plan = mkplan()
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
assert pyka9q.lib().ka9q_hip_init(0) == 0
if shape.P == 1:
    x = bench.make_wideband_torch(shape, plan, nb * shape.L, device)
else:
    x = bench.make_pipelines_torch(shape, plan, nb * shape.L, device)
torch.cuda.synchronize()
bank = bench.build_bank(pyka9q, shape, plan, nb)
if shape.P == 1:
    bank.write_input_device(0, x.data_ptr(), nb * shape.L)
else:
    for i, ch in enumerate(plan):
        bank.write_input_device(i, x[i].data_ptr(), nb * shape.L)
        bank.set_lo(i, ch["lo2"], ch["doppler"], ch["doppler_rate"])
def step():
    for i in range(shape.P):
        bank.commit_input(i, nb * shape.L)
    bank.execute(nb, sync=False)
for _ in range(3):
    step()
bank.sync()
t0 = time.perf_counter()
for _ in range(steps):
    step()
bank.sync()
dt = (time.perf_counter() - t0) / steps
print(f"PLAN={key} channels={len(plan)} blocks={nb} keep_spectrum={bench.KEEP_SPECTRUM}: {dt*1e6:.1f} us/step")
if os.environ.get("PLAN_STAGES"):  # per-stage HIP events, every kernel alone (one stream)
    bank.profile(1)
    for _ in range(5):
        step()
    bank.sync()
    print("stages alone (us):", {k: round(1e3 * ms / n, 1) for k, (ms, n) in bank.stage_times().items() if n})
if hasattr(pyka9q.lib(), "kh_debug_fm2_times"):  # a -DKH_FM2_TIMES build (tools/micro/build_dbg.sh): where kh_fm_demod2's workgroups spend their time
    import ctypes as C, numpy as np
    L_ = pyka9q.lib()
    n = 8192
    buf = np.zeros(n * 8, dtype=np.int64)
    L_.kh_debug_fm2_times.argtypes = [C.c_void_p, C.c_int]
    if L_.kh_debug_fm2_times(buf.ctypes.data, n * 8) == 0:
        tt = buf.reshape(n, 8).astype(np.float64) / 100.0
        tt = tt[tt[:, 3] > 0]
        t0_ = tt[:, 0].min()
        print("kh_fm_demod2 workgroups", len(tt), "span us", round(tt[:, 3].max() - t0_, 2))
        for name, v in (("loads", tt[:, 1] - tt[:, 0]), ("to list", tt[:, 2] - tt[:, 1]), ("items", tt[:, 3] - tt[:, 2]), ("life", tt[:, 3] - tt[:, 0])):
            print(f"  {name:10s} min {v.min():7.2f} p50 {np.median(v):7.2f} p90 {np.percentile(v, 90):7.2f} max {v.max():7.2f}")
bank.close()
