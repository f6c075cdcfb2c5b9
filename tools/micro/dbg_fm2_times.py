#!/usr/bin/env python3
"""Where kh_fm_demod2's time goes (diagnostic build with -DKH_FM2_TIMES, KA9Q_HIP_LIB=.../libka9q_hip_dbg.so):
   per workgroup wall_clock64 stamps at entry, before the SNR iteration, after it, at exit."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, torch, pyka9q
FS, L, M, D, Cn, T = [int(v) for v in sys.argv[1:7]]
torch.cuda.set_device(0)
assert pyka9q.lib().ka9q_hip_init(0) == 0
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(1, T * L, 2, device="cuda", generator=g) * 0.05
t = torch.arange(T * L, device="cuda", dtype=torch.float64)
x[:, :, 0] += 0.3 * torch.cos(2 * np.pi * 0.001 * t).float()
x[:, :, 1] += 0.3 * torch.sin(2 * np.pi * 0.001 * t).float()
bank = pyka9q.Bank(L, M, D, 1, Cn, T, FS, payload=pyka9q.KH_PT_CF32, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
bank.keep_audio(0); bank.keep_baseband(0)
out_rate = FS / D
for c in range(Cn):
    bank.set_channel(c, demod=pyka9q.KH_DEMOD_FM, input=0, rotate=2 * c, low=-0.17 * out_rate, high=0.17 * out_rate, kaiser_beta=3.0)
bank.write_input_device(0, x[0].data_ptr(), T * L)
bank.profile(1)
for _ in range(4):
    bank.commit_input(0, T * L); bank.execute(T, sync=False)
bank.sync()
print({k: round(1e3 * ms / n, 1) for k, (ms, n) in bank.stage_times().items() if n})
L_ = pyka9q.lib()
n = min(8192, Cn)
buf = np.zeros(n * 8, dtype=np.int64)
L_.kh_debug_fm2_times.argtypes = [C.c_void_p, C.c_int]
assert L_.kh_debug_fm2_times(buf.ctypes.data, n * 8) == 0
raw = buf.reshape(n, 8)
tt = raw.astype(np.float64) / 100.0  # us (100 MHz)
keep = tt[:, 3] > 0
tt = tt[keep]; raw = raw[keep]
t0 = tt[:, 0].min()
print("workgroups stamped", len(tt), "span us", tt[:, 3].max() - t0)
for name, v in (("start - first start", tt[:, 0] - t0), ("loads", tt[:, 1] - tt[:, 0]), ("snr", tt[:, 2] - tt[:, 1]), ("passes", tt[:, 3] - tt[:, 2]), ("life", tt[:, 3] - tt[:, 0]), ("end - first start", tt[:, 3] - t0)):
    print(f"{name:22s} min {v.min():7.2f} p50 {np.median(v):7.2f} p90 {np.percentile(v, 90):7.2f} max {v.max():7.2f}")
slow = np.argsort(tt[:, 3] - tt[:, 2])[-3:]
for i in slow:
    print("slow wg", i, "passes", tt[i, 3] - tt[i, 2], "to barrier", tt[i, 4] - tt[i, 2], "outputs", tt[i, 5] - tt[i, 4], "pdev", tt[i, 6] - tt[i, 5], "rest", tt[i, 3] - tt[i, 6], "last it", raw[i, 7])
bank.close()
