#!/usr/bin/env python3
"""step time and per-stage times of the bank for an arbitrary shape (not the contract bench: bench.py is).
   python3 tools/shape_bench.py FS L M D C T [P] [demod: fm|linear|none]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, torch, pyka9q
FS, L, M, D, C, T = [int(v) for v in sys.argv[1:7]]
P = int(sys.argv[7]) if len(sys.argv) > 7 else 1
demod = sys.argv[8] if len(sys.argv) > 8 else "fm"
N = L + M - 1
torch.cuda.set_device(0)
assert pyka9q.lib().ka9q_hip_init(0) == 0
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(P, T * L, 2, device="cuda", generator=g) * 0.05
t = torch.arange(T * L, device="cuda", dtype=torch.float64)
x[:, :, 0] += 0.3 * torch.cos(2 * np.pi * 0.001 * t).float()   # a carrier near DC so that FM squelches open
x[:, :, 1] += 0.3 * torch.sin(2 * np.pi * 0.001 * t).float()
bank = pyka9q.Bank(L, M, D, P, C, T, FS, payload=pyka9q.KH_PT_CF32, compute_n0=int(os.environ.get("SB_N0", "1")), n0_low=-5500.0, n0_high=5500.0)
bank.keep_audio(0)
if demod == "fm":
    bank.keep_baseband(0)
out_rate = FS / D
for c in range(C):
    kw = dict(input=c % P, rotate=(2 * (c // P)) if P == 1 else 0, low=-0.17 * out_rate, high=0.17 * out_rate, kaiser_beta=3.0)
    if demod == "fm":
        bank.set_channel(c, demod=pyka9q.KH_DEMOD_FM, **kw)
    elif demod in ("usb", "usbnoagc", "usbshort"):  # no PLL: the wave-per-channel kernel; usbshort: a hang time of 100 samples
        bank.set_channel(c, demod=pyka9q.KH_DEMOD_LINEAR, pll=0, agc=int(demod != "usbnoagc"),
                         hangtime=100.0 if demod == "usbshort" else 1.1 * out_rate, recovery_rate=10 ** ((6.0 / out_rate) / 20),
                         agc_gain=1e4, **kw)
    elif demod == "linear":
        bank.set_channel(c, demod=pyka9q.KH_DEMOD_LINEAR, pll=1, agc=1, hangtime=1.1 * out_rate, recovery_rate=1.00001, agc_gain=100.0, **kw)
    else:
        bank.set_channel(c, demod=pyka9q.KH_DEMOD_NONE, **kw)
for i in range(P):
    bank.write_input_device(i, x[i].data_ptr(), T * L)
def step():
    for i in range(P):
        bank.commit_input(i, T * L)
    bank.execute(T, sync=False)
for _ in range(3): step()
bank.sync()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
bank.sync()
dt = (time.perf_counter() - t0) / K
bank.profile(1)
for _ in range(5): step()
bank.sync()
st = bank.stage_times()
print(f"FS={FS} L={L} M={M} D={D} C={C} T={T} P={P} {demod}: {dt*1e6:.1f} us/step, {P*T*L/dt/1e6:.0f} MS/s in, realtime x{(T*L/FS)/dt:.1f};",
      {k: round(1e3 * ms / n, 1) for k, (ms, n) in st.items() if n})
bank.close()
