#!/usr/bin/env python3
"""bench workload: distribution of the FM SNR estimate over channels (which blocks take fm_snr's iteration)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, bench, torch, pyka9q
torch.cuda.set_device(0)
assert pyka9q.lib().ka9q_hip_init(0) == 0
plan, rots = bench.channel_plan(0)
x = bench.make_input_torch(plan, torch.device("cuda", 0)); torch.cuda.synchronize()
T, L, M, D, C, FS = bench.T, bench.L, bench.M, bench.D, bench.C, bench.FS
bank = pyka9q.Bank(L, M, D, 1, C, T, FS, payload=pyka9q.KH_PT_CF32, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
for c, r in enumerate(rots):
    bank.set_channel(c, demod=pyka9q.KH_DEMOD_FM, rotate=r, low=-5500.0, high=5500.0, kaiser_beta=3.0)
bank.write_input_device(0, x.data_ptr(), T * L)
for _ in range(3):
    bank.commit_input(0, T * L); bank.execute(T)
snr = np.array([bank.status(c, T)["snr"] for c in range(C)])
print("snr per channel (median over blocks):", np.round(np.median(snr, axis=1), 1))
print("blocks with snr <= 100:", int((snr <= 100).sum()), "of", snr.size, " zero:", int((snr == 0).sum()))
bank.close()
