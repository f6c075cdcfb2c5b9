#!/usr/bin/env python3
"""Randomised parity sweep of the AFSK-1200 / HDLC decoder (kh_afsk_*) against the oracle's restatement of packet.c
(test infrastructure, run on a GPU box): tools/fuzz_afsk.py [cases] [seed].  Each case draws a session count (1..80,
across the 64-lane boundary), a split of the stream into launches, and per session the frames, amplitude, polarity,
noise level (up to where nothing decodes any more), baud-rate error (+-1.5 %: the Gardner steps of packet.c:467-469) and
tone offsets.  Bars as in tests/test_gpu_afsk.py: bit times, frames, RTP counters bit-exact; cur_val 1e-4; baseband 1e-5."""
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

L = orc.AFSK_L


def rel_rms(a, b):
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


def stream(rng, s, nblocks):
    frames = [synth.ax25_ui_frame("APRS", "KA9Q", b"s%d f%d " % (s, k) + bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)),
                                  src_ssid=s % 16) for k in range(int(rng.integers(1, 4)))]
    bits = synth.hdlc_bits(frames, preamble_flags=int(rng.integers(2, 12)))
    x = synth.afsk1200(bits, baud=1200.0 * float(rng.uniform(0.985, 1.015)), amp=float(rng.uniform(0.02, 0.9)),
                       lead=int(rng.integers(0, 4000)), tail=500, phase0=float(rng.uniform(0, 6.28)),
                       mark=1200.0 + float(rng.uniform(-30, 30)), space=2200.0 + float(rng.uniform(-30, 30)))
    n = nblocks * L
    x = x[:n] if x.size > n else np.concatenate([x, np.zeros(n - x.size)])
    x = x + float(rng.choice([0.001, 0.01, 0.05, 0.2])) * rng.standard_normal(n)  # (never digital silence, see the test)
    if rng.random() < 0.3:
        x = -x
    return synth.pcm_be(x), frames


def run_case(rng):
    nsess = int(rng.choice([1, 3, 17, 63, 64, 65, 80]))
    split = [int(v) for v in rng.integers(1, 20, int(rng.integers(1, 5)))]
    nblocks = sum(split)
    xs, sent = zip(*[stream(rng, s, nblocks) for s in range(nsess)])
    dev = hip.Afsk(nsess, max(split), max_frames=8, max_bits=max(split) * 26)
    ors = [orc.AfskSession() for _ in range(nsess)]
    decoded = total = 0
    try:
        pos = 0
        for T in split:
            for s in range(nsess):
                dev.write_pcm(s, xs[s][pos * L:(pos + T) * L])
            dev.execute(T)
            for s in range(nsess):
                yo, fo, bo = [], [], []
                for b in range(T):
                    y, fr, bt = ors[s].block(xs[s][(pos + b) * L:(pos + b + 1) * L])
                    yo.append(y)
                    fo += fr
                    bo += bt
                err = rel_rms(dev.baseband(s, T), np.concatenate(yo))
                assert err < 1e-5, f"session {s} block {pos}: baseband {err:.2e}"
                gb = dev.bits(s)
                assert [int(v) for v in gb["sample"]] == [b[0] for b in bo], f"session {s} block {pos}: bit times differ"
                ref = np.array([b[1] for b in bo], np.float32)
                if ref.size:
                    assert np.abs(gb["cur_val"] - ref).max() <= 1e-4 * np.abs(ref).max(), f"session {s} block {pos}: cur_val"
                assert dev.frames(s) == fo, f"session {s} block {pos}: frames differ"
                decoded += len(fo)
                st = dev.status(s)
                assert (int(st["seq"]), int(st["timestamp"]), int(st["packets"]), int(st["bytes"])) == \
                    (ors[s].st.out_seq, ors[s].st.out_timestamp, ors[s].st.out_packets, ors[s].st.out_bytes), f"session {s}: counters"
            pos += T
        total = sum(len(f) for f in sent)
    finally:
        for o in ors:
            o.close()
        dev.close()
    return nsess, split, decoded, total


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    assert hip.lib().ka9q_hip_init(0) == 0, hip.last_error()
    bad = 0
    t0 = time.time()
    for k in range(cases):
        try:
            nsess, split, decoded, total = run_case(rng)
            print(f"[{k}] ok  sessions={nsess} launches={split}: {decoded} frames decoded on both sides (of {total} sent, some cut off or in heavy noise)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"[{k}] BAD {str(e)[:300]}", flush=True)
    print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
