#!/usr/bin/env python3
"""time per kh_tones_execute launch: python3 tools/tones_bench.py SESSIONS WINDOWS   (not the contract bench)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, pyka9q, synth
S, W = int(sys.argv[1]), int(sys.argv[2])
assert pyka9q.lib().ka9q_hip_init(0) == 0
t = np.arange(W * 2400) / 48000.0
x = 0.1 * np.sin(2 * np.pi * 100.0 * t) + 0.2 * np.sin(2 * np.pi * 770 * t) + 0.2 * np.sin(2 * np.pi * 1336 * t)
pcm = synth.pcm_be(x + 0.01 * np.random.default_rng(0).standard_normal(t.size))
dev = pyka9q.Tones(S, W)
for s in range(S):
    dev.write_pcm(s, pcm)
dev.execute(W)
def step():
    for s in range(S):
        dev.commit_pcm(s, W * 2400)
    dev.execute(W, sync=False)
for _ in range(2): step()
dev.sync()
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
dev.sync()
dt = (time.perf_counter() - t0) / K
print(f"tones: {S} sessions x {W} intervals: {dt*1e3:.3f} ms/launch = {S*W*2400/dt/1e6:.1f} MS/s of 48 kS/s audio, "
      f"{S*(W*0.05)/dt:.0f} sessions in real time; session 0: PL {dev.pl(0)['tone'][:2]}, DTMF {''.join(chr(d) if d else '.' for d in dev.dtmf(0)['digit'][:8])}")
dev.close()
