#!/usr/bin/env python3
"""Randomised parity sweep of the CTCSS / DTMF correlator banks (kh_tones_*) against the oracle's restatement of pl.c
(test infrastructure, run on a GPU box): tools/fuzz_tones.py [cases] [seed].  Each case: 1..70 sessions with a random PL
tone (on or off the table, at random level), random DTMF digits with random timing and twist, voice and noise levels,
polarity, and a random split of the stream into launches.  Bars as in tests/test_gpu_tones.py: decisions exact, energies
1e-4, PL ratio 1e-3 dB, DTMF ratios 1e-3."""
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

W = 2400
LOW, HIGH = [697, 770, 852, 941], [1209, 1336, 1477, 1633]


def audio(rng, n):
    t = np.arange(n) / 48000.0
    x = float(rng.uniform(0, 0.4)) * np.sin(2 * np.pi * float(rng.uniform(300, 3000)) * t + rng.uniform(0, 6.28))
    x += float(rng.choice([0.0005, 0.002, 0.01, 0.05, 0.2])) * rng.standard_normal(n)
    if rng.random() < 0.8:
        hz = float(rng.choice(orc.pl_tones())) if rng.random() < 0.7 else float(rng.uniform(60, 260))
        x += float(rng.uniform(0.005, 0.2)) * np.sin(2 * np.pi * hz * t + rng.uniform(0, 6.28))
    for _ in range(int(rng.integers(0, 4))):
        a = float(rng.uniform(0, n / 48000.0))
        m = (t >= a) & (t < a + float(rng.uniform(0.03, 0.4)))
        lo, hi = LOW[int(rng.integers(4))] * float(rng.uniform(0.985, 1.015)), HIGH[int(rng.integers(4))] * float(rng.uniform(0.985, 1.015))
        x[m] += float(rng.uniform(0.02, 0.3)) * np.sin(2 * np.pi * lo * t[m]) + float(rng.uniform(0.02, 0.3)) * np.sin(2 * np.pi * hi * t[m])
    if rng.random() < 0.3:
        x = -x
    return synth.pcm_be(np.clip(x, -1, 1))


def run_case(rng):
    S = int(rng.choice([1, 5, 33, 64, 70]))
    split = [int(v) for v in rng.integers(1, 12, int(rng.integers(1, 6)))]
    nwin = sum(split)
    xs = [audio(rng, nwin * W) for _ in range(S)]
    dev = hip.Tones(S, max(split))
    ors = [orc.PlSessionRunner() for _ in range(S)]
    npl = ndt = tones = digits = 0
    try:
        pos = 0
        for nw in split:
            for s in range(S):
                dev.write_pcm(s, xs[s][pos * W:(pos + nw) * W])
            dev.execute(nw)
            for s in range(S):
                pl, dt = ors[s].samples(xs[s][pos * W:(pos + nw) * W])
                gp, gd = dev.pl(s), dev.dtmf(s)
                assert len(gp) == len(pl) and len(gd) == len(dt) == nw, f"session {s} window {pos}: result counts"
                for a, b in zip(gp, pl):
                    assert a["tone"] == np.float32(b["tone"]), f"session {s} window {pos}: PL tone {a} vs {b}"
                    tones += b["tone"] != 0
                    if np.isnan(b["snr"]):
                        assert np.isnan(a["snr"])
                        continue
                    assert a["index"] == b["index"], f"session {s} window {pos}: PL index {a} vs {b}"
                    assert abs(a["e0"] - b["e0"]) <= 1e-4 * b["e0"] and abs(a["e1"] - b["e1"]) <= 1e-4 * b["e1"], f"session {s}: PL energies {a} vs {b}"
                    assert abs(a["snr"] - b["snr"]) <= 1e-3, f"session {s}: PL ratio {a} vs {b}"
                for a, b in zip(gd, dt):
                    assert (chr(a["digit"]) if a["digit"] else "") == b["digit"], f"session {s} window {pos}: digit {a} vs {b}"
                    digits += b["digit"] != ""
                    assert (a["low_index"], a["high_index"]) == (b["low_index"], b["high_index"]), f"session {s}: DTMF indices {a} vs {b}"
                    for k in ("low_snr", "high_snr"):
                        # pl.c:395-410: snr = max / (total - max).  The subtraction cancels: one fp32 rounding of either energy
                        # (6e-8) moves the quotient by (snr + 1) times that, on the reference's side as on the GPU's -- at
                        # snr 2e4 the reference's own value is only good to 2e-3.  The bar follows the formula's condition
                        # number (two roundings a side); it is 1e-3 up to snr ~ 1e3, where the digits are decided (snr >= 10).
                        tol = 1e-3 + 2.5e-7 * abs(b[k])
                        assert (np.isnan(a[k]) and np.isnan(b[k])) or abs(a[k] - b[k]) <= tol * abs(b[k]), f"session {s}: {k} {a} vs {b}"
                npl += len(pl)
                ndt += len(dt)
            pos += nw
    finally:
        dev.close()
    return S, split, npl, ndt, int(tones), int(digits)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    assert hip.lib().ka9q_hip_init(0) == 0, hip.last_error()
    bad = 0
    tot = np.zeros(4, np.int64)
    t0 = time.time()
    for k in range(cases):
        try:
            S, split, *cnt = run_case(rng)
            tot += cnt
            print(f"[{k}] ok  sessions={S} launches={split}: {cnt[0]} PL decisions ({cnt[2]} with a tone), {cnt[1]} DTMF decisions ({cnt[3]} with a digit)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"[{k}] BAD {str(e)[:500]}", flush=True)
    print(f"{cases} cases, {bad} bad; {tot[0]} PL decisions ({tot[2]} with a tone), {tot[1]} DTMF decisions ({tot[3]} with a digit); {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
