#!/usr/bin/env python3
"""Randomised parity sweep of the bank's ingest stage -- wire format, gain_factor, LO2 / Doppler oscillators switched on,
retuned and off again between launches, zero-filled gaps, overlap-save history across launches -- against the oracle
pipeline (test infrastructure, run on a GPU box): tools/fuzz_ingest.py [cases] [seed].  Shapes cover the single-workgroup
transforms (N <= 8192, separate kh_ingest) and the four-step transform whose column pass reads the raw ring itself,
including L < M-1 (the history feeds the next history).  Bars as in tests/test_gpu_ingest_decimate.py: spectrum and
baseband 1e-5 relative RMS, IF power 1e-5."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import orc  # noqa: E402
import pipeline  # noqa: E402
import pyka9q as hip  # noqa: E402
import synth  # noqa: E402
from test_gpu_ingest_decimate import _raw  # noqa: E402

SHAPES = [(3840, 4353, 4), (3840, 3841, 4), (512, 513, 2), (8192, 8193, 16), (4096, 12289, 16), (16384, 16385, 32),
          (32768, 32769, 64), (12288, 20481, 32), (65536, 65537, 128)]


def rel_rms(a, b):
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


def run_case(rng, sums):
    L, M, D = SHAPES[int(rng.integers(len(SHAPES)))]
    FS = 48_000 * D
    fill = -(-(M - 1) // L)  # blocks until the overlap-save history holds real input (before that the kept outputs are
    # the filter's start-up transient, far below the transform's intermediate values: see tools/fuzz_filter.py)
    payload = int(rng.choice([0, orc.IQ_PT, orc.PCM_STEREO_PT, orc.IQ_PT8, orc.IQ_PT12]))
    T, calls = int(rng.integers(1, 5)), int(rng.integers(2, 6))
    while T * calls * L > 1_500_000 and calls > 2:
        calls -= 1
    gain = float(rng.choice([1.0, 0.75, 3.3]))
    tone = float(rng.uniform(-3e3, 3e3))
    x = (synth.noise(calls * T * L, 0.05, int(rng.integers(1, 100)))
         + synth.fm_carrier(0, calls * T * L, FS, tone, 0.3, 700.0, 1000.0)).astype(np.complex64)
    raw = x if payload == 0 else _raw(payload, x, rng)
    per = raw.size // (calls * T)
    bank = hip.Bank(L, M, D, 1, 1, T, FS, payload=payload)
    pipe = pipeline.Pipeline(L, M, D, FS, payload=payload)
    bank.set_gain(0, gain)
    pipe.fins[0].gain = gain
    bank.set_channel(0, low=-4000.0, high=4000.0)
    pipe.add_channel(low=-4000.0, high=4000.0)
    olen = L // D
    level = gain * float(np.sqrt(np.mean(np.abs(x) ** 2)))  # (12-bit payloads are random bytes: larger, so this is a floor)
    log = []
    worst = 0.0
    try:
        for call in range(calls):
            ev = rng.random()
            if ev < 0.25:  # both oscillators (a sweep on the Doppler one)
                lo2, df, dr = float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-1e-9, 1e-9))
                bank.set_lo(0, lo2, df, dr)
                pipe.set_lo(0, lo2, df, dr)
                log.append(f"{call}:lo")
            elif ev < 0.4:  # retune LO2 only, phase kept (radio.c:350, osc.c:24-27)
                lo2 = float(rng.uniform(-0.1, 0.1))
                bank.set_second_lo(0, lo2)
                orc.lib().orc_set_osc(C.addressof(pipe.fins[0].lo2), lo2, 0.0)
                log.append(f"{call}:retune")
            elif ev < 0.5:  # everything off again: back to the fused path where there is one
                bank.set_lo(0, 0.0, 0.0, 0.0)
                pipe.set_lo(0, 0.0, 0.0, 0.0)
                log.append(f"{call}:off")
            gap = int(rng.integers(T)) if rng.random() < 0.2 else -1  # one lost block (radio.c:112-131)
            seg = raw[call * T * per:(call + 1) * T * per]
            for b in range(T):
                if b == gap:
                    bank.write_zeros(0, L)
                else:
                    bank.write_input(0, seg[b * per:(b + 1) * per], count=L)
            bank.execute(T)
            bb = bank.baseband(0, T)
            ist = bank.input_status(0, T)
            for b in range(T):
                blk = seg[b * per:(b + 1) * per]
                if b == gap:
                    blk = np.zeros_like(blk)
                    if payload == orc.IQ_PT12:
                        pass  # twelve-bit zero is all-zero bytes as well
                outs, istat = pipe.run_block([blk])
                e1 = rel_rms(bank.spectrum(0, b), pipe.spectrum(0))
                e2 = rel_rms(bb[b * olen:(b + 1) * olen], outs[0][0])
                worst = max(worst, e1)
                assert e1 < 1e-5, f"spectrum call {call} block {b}: {e1:.2e} ({log})"
                # a block whose output is far below its input (right after a gap, or while the history fills) is a
                # transient: its relative error measures the transform's rounding floor, not the path
                xin = pipe.convert(0, blk)
                transient = call * T + b < fill or np.linalg.norm(outs[0][0]) / np.sqrt(olen) < 0.05 * level
                assert e2 < 1e-5 or transient, (f"baseband call {call} block {b}: {e2:.2e} gap={gap} ({log}) L={L} M={M} "
                                                f"payload={payload} T={T} tone={tone:.0f}")
                if not transient:
                    worst = max(worst, e2)
                g, o = float(ist[b]["if_power"]), float(istat[0][0])
                if abs(g - o) > 1e-5 * o + 1e-12:
                    # radio.c:188 adds L squared magnitudes into one fp32 variable; from L = 8192 on that running sum
                    # is itself 1e-5..4e-5 off.  Yardstick: the same mean in float64 (the oscillators have unit modulus).
                    t = float(np.mean(np.abs(xin.astype(np.complex128)) ** 2))
                    assert abs(g - t) <= 3e-6 * t and abs(o - t) <= 1e-4 * t, \
                        f"if_power call {call} block {b}: gpu {g} oracle {o} float64 {t} gap={gap} ({log}) L={L} payload={payload}"
                    sums.append((L, abs(g - t) / t, abs(o - t) / t))
    finally:
        bank.close()
    return f"L={L} M={M} D={D} payload={payload} T={T} calls={calls} gain={gain} events={log}", worst


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    assert hip.lib().ka9q_hip_init(0) == 0, hip.last_error()
    bad = 0
    sums = []
    t0 = time.time()
    for k in range(cases):
        try:
            tag, worst = run_case(rng, sums)
            print(f"[{k}] ok  {worst:.2e} {tag}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"[{k}] BAD {str(e)[:400]}", flush=True)
    for t in sums:
        print("if_power beyond 1e-5 of the oracle: L=%d, gpu within %.1e of float64, the fp32 running sum within %.1e" % t)
    print(f"{cases} cases, {bad} bad, {len(sums)} blocks with the reference's IF-power running sum off, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
