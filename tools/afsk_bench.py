#!/usr/bin/env python3
"""time per kh_afsk_execute launch: python3 tools/afsk_bench.py SESSIONS BLOCKS   (not the contract bench)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, pyka9q, synth
S, T = int(sys.argv[1]), int(sys.argv[2])
assert pyka9q.lib().ka9q_hip_init(0) == 0
frames = [synth.ax25_ui_frame("APRS", "KA9Q", b"bench frame %d" % k) for k in range(3)]
x = synth.afsk1200(synth.hdlc_bits(frames), amp=0.4, lead=500, tail=500)
x = np.resize(x, T * 1000) + 0.003 * np.random.default_rng(0).standard_normal(T * 1000)
pcm = synth.pcm_be(x)
dev = pyka9q.Afsk(S, T, max_frames=8)
ptr, stride = dev.pcm_device_ptr()
for s in range(S):
    dev.write_pcm(s, pcm)
def step():
    for s in range(S):
        dev.commit_pcm(s, T * 1000)   # the rows keep their contents: same audio again
    dev.execute(T, sync=False)
dev.execute(T)
for _ in range(2): step()
dev.sync()
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
dev.sync()
dt = (time.perf_counter() - t0) / K
n = sum(len(dev.frames(s)) for s in range(min(S, 4)))
print(f"AFSK: {S} sessions x {T} blocks: {dt*1e3:.3f} ms/launch = {S*T*1000/dt/1e6:.1f} MS/s of 48 kS/s audio, "
      f"{S*(T*1000/48000)/dt:.0f} sessions in real time; frames in the first sessions' last launch: {n}")
dev.close()
