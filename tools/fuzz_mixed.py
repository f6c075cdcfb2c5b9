#!/usr/bin/env python3
"""Randomised parity sweep of banks that mix demodulators (test infrastructure, run on a GPU box):
tools/fuzz_mixed.py [cases] [seed].  Each case: 4 inputs, 1..24 channels, every channel with a random input and a random
demodulator (FM with or without de-emphasis, linear.c without PLL in the variants of tools/fuzz_linear.py, am.c), random
block and launch counts -- the bank sorts channels into per-demodulator lists and keeps per-channel state across
launches, which is what this exercises.  Bars: PCM +-1 LSB, audio 2e-4 of the block's peak (FM: rtol 1e-4), status 1e-4
(FM snr with its 1e-4 floor)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import orc  # noqa: E402
import pipeline  # noqa: E402
import pyka9q as hip  # noqa: E402
from fuzz_linear import SIGS, draw_channel, is_agc_tie, tie_goes_on, pcm_follows_audio, Replay  # noqa: E402
from test_gpu_bank_linear import FS, L, M, D, OLEN, _signal  # noqa: E402


def run_case(rng, ties):
    T, calls = int(rng.integers(1, 6)), int(rng.integers(1, 5))
    C = int(rng.integers(1, 25))
    n = T * calls * L
    chans = []
    for _ in range(C):
        if rng.random() < 0.4:
            chans.append(dict(demod="fm", input=int(rng.integers(4)), low=-float(rng.uniform(3000, 8000)), high=float(rng.uniform(3000, 8000)),
                              flat=int(rng.integers(2)), headroom=float(10 ** (-rng.uniform(3, 25) / 20))))
        else:
            chans.append(draw_channel(rng))
    sigs = {s: _signal(s, n).astype(np.complex64) for s in SIGS}
    bank = hip.Bank(L, M, D, 4, C, T, FS)
    pipe = pipeline.Pipeline(L, M, D, FS, n_inputs=4)
    ids = {"linear": hip.KH_DEMOD_LINEAR, "am": hip.KH_DEMOD_AM, "fm": hip.KH_DEMOD_FM}
    for c, kw in enumerate(chans):
        kw = dict(kw)
        dem = kw.pop("demod")
        bank.set_channel(c, demod=ids[dem], **kw)
        pipe.add_channel(demod=dem, **kw)
    hist = []  # the oracle's status records, block by block (for the diagnostics below)
    open_tie = {}  # channels inside a divergence that began with a verified AGC branch tie (tools/fuzz_linear.py)
    replay = Replay(chans, int(np.float32(FS) / np.float32(D)))  # the reference loop over the GPU's own baseband (tools/fuzz_linear.py)
    try:
        for call in range(calls):
            for i, s in enumerate(SIGS):
                bank.write_input(i, sigs[s][call * T * L:(call + 1) * T * L])
            bank.execute(T)
            g_pcm = [bank.pcm(c, T) for c in range(C)]
            g_aud = [bank.audio(c, T) for c in range(C)]
            g_st = [bank.status(c, T) for c in range(C)]
            replay.run(bank, T, OLEN)
            for b in range(T):
                lo = (call * T + b) * L
                outs, _ = pipe.run_block([sigs[s][lo:lo + L] for s in SIGS])
                hist.append([o[2] for o in outs])
                for c, kw in enumerate(chans):
                    _y, audio, sig = outs[c]
                    what = f"ch {c}/{C} {kw} block {b} call {call}"
                    nch = kw.get("channels", 1) if kw["demod"] == "linear" and not kw.get("env") else 1
                    sl = slice(b * OLEN * nch, (b + 1) * OLEN * nch)
                    ga, gp = g_aud[c][sl], g_pcm[c][sl]
                    if call == 0 and b == 0 and kw["demod"] != "fm":  # start-up on amplified round-off (see the fixed test)
                        ga, gp, audio = ga[600 * nch:], gp[600 * nch:], audio[600 * nch:]
                    d = np.abs(gp.astype(np.int32) - orc.scaleclip(audio).astype(np.int32))
                    st = g_st[c][b]
                    if kw["demod"] == "fm":
                        if not np.allclose(ga, audio, rtol=1e-4, atol=2e-5):
                            e = np.abs(ga - audio) - 1e-4 * np.abs(audio)
                            i = int(np.argmax(e))
                            # fm.c:86,123: after a closed block state = 0, and sample 0 of the next open block is
                            # cargf(y[0] * conjf(0)): a pair of signed zeros, for which atan2f answers 0, -0, pi or -pi by the
                            # signs of y[0]'s components.  When one component of y[0] is itself zero to rounding (here the two
                            # sides may hold +-1e-9 of a 0.05 sample) the answer is a coin toss in the reference too.  Verified,
                            # not assumed: only sample 0 is off, both values are 0 or +-pi*gain, and a component of y[0] is
                            # below 1e-6 of its magnitude on both sides.
                            gain_fm = kw["headroom"] / np.pi * (FS / D) / abs(kw["low"] - kw["high"])
                            y0g, y0o = complex(bank.baseband(c, T)[b * OLEN]), complex(_y[0])
                            knife = all(min(abs(v.real), abs(v.imag)) <= 1e-6 * abs(v) for v in (y0g, y0o))
                            quant = all(min(abs(abs(v) - q) for q in (0.0, np.pi * gain_fm)) <= 1e-5 for v in (float(ga[0]), float(audio[0])))
                            if i == 0 and int((e > 2e-5).sum()) == 1 and kw.get("flat") and knife and quant:
                                ties.append(f"{what}: signed-zero tie at sample 0 (y[0] gpu {y0g!r} oracle {y0o!r})")
                                continue
                            raise AssertionError(f"{what}: FM audio, worst at sample {i} of {ga.size}: gpu "
                                                 f"{np.array2string(ga[max(i - 2, 0):i + 3], precision=7)} oracle "
                                                 f"{np.array2string(audio[max(i - 2, 0):i + 3], precision=7)}; snr gpu {st['snr']:.6g} "
                                                 f"oracle {sig['snr']:.6g}, bb_power {st['bb_power']:.6g} / {sig['bb_power']:.6g}, "
                                                 f"samples beyond the bar: {int((e > 2e-5).sum())}; y[0] gpu {bank.baseband(c, T)[b * OLEN]!r} oracle {_y[0]!r}; oracle (snr, squelch) of the blocks so far "
                                                 f"{[(round(float(h[c]['snr']), 3), int(h[c].get('squelch_open', -1))) for h in hist[-4:]]}, gpu "
                                                 f"{[(round(float(g_st[c][bb]['snr']), 3), int(g_st[c][bb]['squelch_open'])) for bb in range(b + 1)]}")
                        assert d.max() <= 1, f"{what}: FM pcm max diff {d.max()}"
                        for key in ("bb_power", "snr", "output_level"):
                            floor = 1e-4 if key == "snr" else 1e-12
                            assert abs(st[key] - sig[key]) <= 1e-4 * abs(sig[key]) + floor, f"{what}: {key} {st[key]} vs {sig[key]}"
                        continue
                    scale = max(float(np.abs(audio).max()), 1e-6)
                    err = float(np.abs(ga - audio).max()) / scale
                    skip = 600 * nch if call == 0 and b == 0 else 0
                    if kw["demod"] == "linear" and kw.get("agc") and (err > 2e-4 or d.max() > 1 or abs(st["agc_gain"] - sig["agc_gain"]) > 1e-4 * abs(sig["agc_gain"])) \
                            and replay.confirms(c, b, g_aud[c][sl], g_pcm[c][sl], st, skip):
                        ties.append(f"{what}: audio err {err:.2e} against the oracle; the reference loop over the GPU's own baseband gives the GPU's output")
                        continue
                    if err > 2e-4:
                        rel = np.abs(ga - audio) / scale
                        i = int(np.argmax(rel > 5e-5))
                        # the hang AGC's branch tie (tools/fuzz_linear.py), with a hang time: the side that took the attack
                        # branch holds its gain for hangtime samples while the other recovers
                        tie = bool(kw.get("agc")) and (is_agc_tie(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch) or
                                                        bool(open_tie.get(c) and tie_goes_on(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch)))
                        msg = (f"{what}: audio err {err:.2e}, first beyond 5e-5 at sample {i} of {ga.size}: gpu "
                               f"{np.array2string(ga[max(i - 3, 0):i + 4], precision=7)} oracle "
                               f"{np.array2string(audio[max(i - 3, 0):i + 4], precision=7)}; rel err there "
                               f"{np.array2string(rel[max(i - 3, 0):i + 4], precision=2)}, at the end {rel[-1]:.2e}")
                        if tie:
                            ties.append(msg)
                            open_tie[c] = True
                            continue
                        raise AssertionError(msg + "; tie test: " + str(getattr(is_agc_tie, "why", "")))
                    open_tie[c] = False
                    if d.max() > 1 and kw.get("agc") and is_agc_tie(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch):
                        ties.append(what)  # AGC branch tie (tools/fuzz_linear.py)
                        continue
                    over = np.flatnonzero(d > 1)
                    if over.size and pcm_follows_audio(ga, gp, audio, over):
                        ties.append(f"{what}: PCM exact for the GPU's own audio, block over full scale (tools/fuzz_linear.py)")
                        continue
                    if d.max() == 2 and all(abs(abs(float(audio[j])) - 1.0) <= 2e-7 and abs(abs(float(ga[j])) - 1.0) <= 2e-7 for j in over):
                        ties.append(f"{what}: clip boundary of scaleclip (audio.c:22-28: -32766 -> -32768 at x = -1)")
                        continue
                    assert d.max() <= 1, (f"{what}: pcm max diff {d.max()}, block audio err {err:.2e}; audio there gpu "
                                          f"{[float(ga[j]) for j in over[:4]]} oracle {[float(audio[j]) for j in over[:4]]}")
                    keys = ["bb_power", "agc_gain"] if kw["demod"] == "am" else ["bb_power", "agc_gain", "output_level"]
                    for key in keys:
                        if key == "output_level" and call == 0 and b == 0:  # (start-up samples count in it: against the GPU's own audio, tools/fuzz_linear.py)
                            full = np.asarray(g_aud[c][sl], np.float64)
                            own = float((full * full).sum() / (OLEN * nch))
                            assert abs(st[key] - own) <= 1e-5 * abs(own) + 1e-12, f"{what}: {key} {st[key]} vs own audio {own}"
                            continue
                        assert abs(st[key] - sig[key]) <= 1e-4 * abs(sig[key]) + 1e-12, f"{what}: {key} {st[key]} vs {sig[key]}"
    finally:
        bank.close()
    return f"T={T} calls={calls} channels={C} ({sum(k['demod'] == 'fm' for k in chans)} FM, {sum(k['demod'] == 'linear' for k in chans)} linear, {sum(k['demod'] == 'am' for k in chans)} AM)"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    assert hip.lib().ka9q_hip_init(0) == 0, hip.last_error()
    bad = 0
    ties = []
    t0 = time.time()
    for k in range(cases):
        try:
            print(f"[{k}] ok  {run_case(rng, ties)}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"[{k}] BAD {str(e)[:2400]}", flush=True)
    for t in ties:
        print("agc tie:", t)
    print(f"{cases} cases, {bad} bad, {len(ties)} blocks with a verified branch tie (AGC attack, signed zero at a squelch opening, or scaleclip's clip boundary), {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
