#!/usr/bin/env python3
"""Randomised parity sweep of the RTP packetisation plan (kh_rtp_plan: audio.c:32-135 on the device) against the oracle's
packetiser run over the bank's own PCM (test infrastructure, run on a GPU box): tools/fuzz_rtp.py [cases] [seed].  Random
keying of an FM carrier (so that whole chunks are silent), a stereo linear channel, random block/launch counts, two bank
shapes (olen 960 and 480), random starting timestamp / sequence number / silent flag (including values that wrap inside
the run).  Integer logic: everything bit-exact."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import orc  # noqa: E402
import pyka9q as hip  # noqa: E402
import synth  # noqa: E402


def run_case(rng):
    fs, L, M, D = [(192_000, 3840, 4353, 4), (192_000, 3840, 4353, 8), (96_000, 1920, 2177, 2)][int(rng.integers(3))]
    olen = L // D
    T, calls = int(rng.integers(1, 8)), int(rng.integers(1, 4))
    n = T * calls * L
    keyed = (rng.random(T * calls) < rng.uniform(0.2, 0.8)).astype(np.float32)
    x = (synth.noise(n, 0.0, 5) + np.repeat(keyed, L) * synth.fm_carrier(0, n, fs, 0.0, 0.4, 700.0, 2500.0)).astype(np.complex64)
    out_rate = fs / D
    edge = min(5500.0, 0.45 * out_rate)
    bank = hip.Bank(L, M, D, 1, 3, T, fs)
    bank.rtp_enable(1)
    bank.set_channel(0, demod=hip.KH_DEMOD_FM, rotate=0, low=-edge, high=edge, kaiser_beta=3.0)
    bank.set_channel(1, demod=hip.KH_DEMOD_LINEAR, rotate=0, low=-edge, high=edge, kaiser_beta=3.0, channels=2, agc=0, agc_gain=1.0)
    bank.set_channel(2, rotate=0, low=-edge, high=edge)
    ts0 = int(rng.choice([0, 0xFFFFF000, int(rng.integers(0, 2 ** 32))]))
    seq0 = int(rng.choice([0, 0xFFFE, int(rng.integers(0, 65536))]))
    sil0 = int(rng.integers(2))
    bank.rtp_set_state(0, timestamp=ts0, seq=seq0, silent=sil0)
    ref = [orc.RtpPacketiser(1, ts0, seq0, sil0), orc.RtpPacketiser(2)]
    sent = supp = 0
    try:
        for call in range(calls):
            bank.write_input(0, x[call * T * L:(call + 1) * T * L])
            bank.execute(T)
            for ch, chans in ((0, 1), (1, 2)):
                pcm = bank.pcm(ch, T)
                plan, after = bank.rtp(ch)
                wpb = olen * chans
                want = []
                for b in range(T):
                    want += ref[ch].block(pcm[b * wpb:(b + 1) * wpb], base_offset=b * wpb)
                assert len(plan) == len(want) == T * ((wpb + 479) // 480), f"channel {ch} call {call}: chunk count"
                for k, (got, w) in enumerate(zip(plan, want)):
                    ts, seq, marker, send, off, nw = w
                    assert (int(got["timestamp"]), int(got["send"]), int(got["offset"]), int(got["nwords"])) == (ts, send, off, nw), \
                        f"channel {ch} call {call} chunk {k}: {got} vs {w}"
                    if send:
                        assert (int(got["seq"]), int(got["marker"])) == (seq, marker), f"channel {ch} call {call} chunk {k}: {got} vs {w}"
                r = ref[ch]
                assert (int(after["timestamp"]), int(after["seq"]), int(after["silent"]), int(after["packets"]), int(after["bytes"]),
                        int(after["samples"])) == (r.timestamp, r.seq, r.silent, r.packets, r.bytes, r.samples), f"channel {ch} call {call}: state"
                if ch == 0:
                    sent += int(plan["send"].sum())
                    supp += int((plan["send"] == 0).sum())
            assert len(bank.rtp(2)[0]) == 0
    finally:
        bank.close()
    return f"olen={olen} T={T} calls={calls} keyed={int(keyed.sum())}/{keyed.size}: FM channel sent {sent}, suppressed {supp} chunks"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    assert hip.lib().ka9q_hip_init(0) == 0, hip.last_error()
    bad = 0
    t0 = time.time()
    for k in range(cases):
        try:
            print(f"[{k}] ok  {run_case(rng)}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"[{k}] BAD {str(e)[:500]}", flush=True)
    print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
