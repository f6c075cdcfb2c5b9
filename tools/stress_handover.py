#!/usr/bin/env python3
"""Hand-overs between the bank's streams under load: bursts of K launches queued without a host sync (K = 3 .. 11), each
launch fed a DIFFERENT signal (the carriers move from launch to launch, so a back half that read a stale spectrum slot --
the one three launches old -- or a front half that overwrote a slot too early would demodulate other tones), against the
same sequence on a bank whose launches run on one stream with a sync after each (KA9Q_HIP_PIPELINE=0: no hand-over
events at all).  After every burst: PCM of every channel, per-channel status and input status of the burst's last launch,
bit for bit.  What this guards: the events that order the two streams carry no system-scope fence
(hipEventDisableSystemFence) and the host waits for a slot's last reader (KA9Q_HIP_SLOT_HOSTWAIT).

usage: tools/stress_handover.py [bursts] [seed] [mixed]   (mixed: the last four channels are USB with hang AGC and envelope AM --
their demodulators run on a third stream beside the FM demodulator, forked and joined by two more such events)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np
import pyka9q as hip, synth

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"
rng = np.random.default_rng(seed)
fs, L, M, D, C, T = 8_000_000, 131072, 131073, 256, 16, 4
N = L + M - 1
assert hip.lib().ka9q_hip_init(0) == 0
# a pool of launch inputs; in variant v the carrier of channel c sits in slot (c + v) % C with its own tone
ks = [k - C // 2 for k in range(C)]
NV = 5
pool = []
for v in range(NV):
    sigs = [(ks[(c + v) % C], 0.02 + 0.004 * c, 300.0 + 45 * c + 7 * v, 2500.0) for c in range(0, C, 2)]  # every other slot busy
    pool.append(np.ascontiguousarray(synth.wideband_fm(100 + v, T * L, fs, N, sigs, stream=v), np.complex64))


def make(env):
    for k, val in env.items():
        os.environ[k] = val
    b = hip.Bank(L, M, D, 1, C, T, fs, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
    out_rate = fs / D
    for c, k in enumerate(ks):
        rot = synth.rotate_for(k, 12500.0, N, fs)
        if mixed and c >= C - 4:
            b.set_channel(c, demod=hip.KH_DEMOD_LINEAR, rotate=rot, low=100.0 if c % 2 else -5000.0, high=3000.0 if c % 2 else 5000.0,
                          kaiser_beta=3.0, pll=0, agc=1, env=int(c % 2 == 0), hangtime=1.1 * out_rate,
                          recovery_rate=10 ** ((6.0 / out_rate) / 20), agc_gain=1e3)
        else:
            b.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=rot, low=-5500.0, high=5500.0, kaiser_beta=3.0)
    for k in env:
        del os.environ[k]
    return b


ref = make({"KA9Q_HIP_PIPELINE": "0"})
dut = make({})
src = hip.Bank(L, M, D, 1, 1, T * NV, fs)  # only its raw ring is used: the pool, resident on the device
src.write_input(0, np.concatenate(pool))
xd, _ = src.raw_device_ptr(0)
t0 = time.time()
bad = 0
launches = 0
for burst in range(bursts):
    K = int(rng.integers(3, 12))
    seq = [int(rng.integers(0, NV)) for _ in range(K)]
    nbs = [int(rng.integers(1, T + 1)) for _ in range(K)]
    for v, nb in zip(seq, nbs):
        ref.write_input(0, pool[v][:nb * L])
        ref.execute(nb)
        ref.sync()
    for v, nb in zip(seq, nbs):
        dut.write_input_device(0, xd + 8 * v * T * L, nb * L)
        dut.execute(nb, sync=False)
    launches += K
    nb = nbs[-1]
    ok = np.array_equal(ref.pcm_all(nb), dut.pcm_all(nb))
    ok = ok and all(ref.status(c, nb).tobytes() == dut.status(c, nb).tobytes() for c in range(C))
    ok = ok and ref.input_status(0, nb).tobytes() == dut.input_status(0, nb).tobytes()
    if not ok:
        bad += 1
        print(f"burst {burst}: K={K} variants {seq} blocks {nbs}: DIFFERENT", flush=True)
    if burst % 50 == 49:
        print(f"{burst + 1} bursts, {launches} launches, {bad} bad, {time.time() - t0:.0f} s", flush=True)
assert np.abs(dut.pcm_all(nb)).max() > 100
print(f"stress_handover: {bursts} bursts, {launches} launches, {bad} bad, {time.time() - t0:.0f} s")
for b in (ref, dut, src):
    b.close()
sys.exit(1 if bad else 0)
