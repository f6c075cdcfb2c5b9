#!/usr/bin/env python3
"""Randomised parity sweep of the bank's linear.c / am.c demodulators against the oracle pipeline (test
infrastructure, run on a GPU box): tools/fuzz_linear.py [cases] [seed] [decimation 4|8|16].  Each case is a bank of 6 channels with random
mode parameters (filter edges, AGC on/off, hang time, recovery rate, start gain, headroom, mono/stereo, ISB fold,
envelope detection, frequency shift, am.c) over the four test signals of tests/test_gpu_bank_linear.py, random block and
launch counts; the bars are that test's (audio 2e-4 of the block's peak, PCM +-1 LSB, status 1e-4).  The PLL modes
stay with the fixed test: their first blocks start on amplified round-off and need its settling allowance."""
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
from test_gpu_bank_linear import FS, L, M, _signal  # noqa: E402

D = 4  # decimation (main() takes 4, 8 or 16 from the command line: olen 960, 480, 240, kh_demod_linear_w's three tilings)
OLEN = L // D

SIGS = ["usb", "am", "dsb", "iq"]


def draw_channel(rng):
    kw = dict(input=int(rng.integers(4)), headroom=float(10 ** (-rng.uniform(3, 25) / 20)))
    if rng.random() < 0.2:
        kw.update(demod="am", low=-5000.0, high=5000.0, hangtime=float(rng.choice([0.0, 0.5, 1.1])),
                  recovery_rate=float(rng.uniform(1, 20)), agc_gain=float("nan"))
        return kw
    ssb = rng.random() < 0.4
    low, high = (float(rng.uniform(50, 400)), float(rng.uniform(2000, 3500))) if ssb else (-float(rng.uniform(2000, 6000)), float(rng.uniform(2000, 6000)))
    if ssb and rng.random() < 0.5:
        low, high = -high, -low  # LSB
    kw.update(demod="linear", low=low, high=high, channels=int(rng.integers(1, 3)))
    if rng.random() < 0.7:
        kw.update(agc=1, hangtime=float(rng.choice([0.0, 100.0, 1.1 * 48000])),
                  recovery_rate=float(10 ** ((rng.uniform(1, 40) / 48000) / 20)), agc_gain=float(10 ** (rng.uniform(0, 80) / 20)))
    else:
        kw.update(agc=0, agc_gain=float(rng.uniform(0.5, 20)))
    if rng.random() < 0.3:
        kw["shift"] = float(rng.uniform(-2000, 2000) / 48000)
    if kw["channels"] == 2 and rng.random() < 0.4:
        kw["out_type"] = orc.CROSS_CONJ  # ISB
    elif rng.random() < 0.25:
        kw.update(env=1, channels=1)
    return kw


def run_case(rng, k, ties):
    T, calls = int(rng.integers(1, 6)), int(rng.integers(1, 5))
    n = T * calls * L
    chans = [draw_channel(rng) for _ in range(6)]
    sigs = {s: _signal(s, n).astype(np.complex64) for s in SIGS}
    bank = hip.Bank(L, M, D, 4, len(chans), T, FS)
    pipe = pipeline.Pipeline(L, M, D, FS, n_inputs=4)
    ids = {"linear": hip.KH_DEMOD_LINEAR, "am": hip.KH_DEMOD_AM}
    for c, kw in enumerate(chans):
        kw = dict(kw)
        dem = kw.pop("demod")
        bank.set_channel(c, demod=ids[dem], **kw)
        pipe.add_channel(demod=dem, **kw)
    worst = 0.0
    replay = Replay(chans, int(np.float32(FS) / np.float32(D)))
    open_tie = {}  # channels inside a divergence that began with a verified branch tie
    try:
        for call in range(calls):
            for i, s in enumerate(SIGS):
                bank.write_input(i, sigs[s][call * T * L:(call + 1) * T * L])
            bank.execute(T)
            g_pcm = [bank.pcm(c, T) for c in range(len(chans))]
            g_aud = [bank.audio(c, T) for c in range(len(chans))]
            g_st = [bank.status(c, T) for c in range(len(chans))]
            replay.run(bank, T, OLEN)
            if os.environ.get("KA9Q_FUZZ_DUMP"):  # digests of the GPU's outputs: two builds of the library can be compared bit for bit
                import zlib
                with open(os.environ["KA9Q_FUZZ_DUMP"], "a") as f:
                    for c in range(len(chans)):
                        f.write(f"{k} {call} {c} {zlib.crc32(g_aud[c].tobytes()):08x} {zlib.crc32(g_pcm[c].tobytes()):08x} {zlib.crc32(g_st[c].tobytes()):08x}\n")
            for b in range(T):
                lo = (call * T + b) * L
                outs, _ = pipe.run_block([sigs[s][lo:lo + L] for s in SIGS])
                for c, kw in enumerate(chans):
                    _y, audio, sig = outs[c]
                    nch = kw.get("channels", 1) if kw["demod"] == "linear" and not kw.get("env") else 1
                    sl = slice(b * OLEN * nch, (b + 1) * OLEN * nch)
                    ga, gp = g_aud[c][sl], g_pcm[c][sl]
                    if call == 0 and b == 0:  # start-up on amplified round-off (see the fixed test)
                        ga, gp, audio = ga[(2400 // D) * nch:], gp[(2400 // D) * nch:], audio[(2400 // D) * nch:]
                    if kw["demod"] == "am":
                        # am.c:56-59,76-85: the same attack / hang / recover AGC on the carrier estimate, with the rate given
                        # in dB/s and the hang in seconds (times the output rate: am.c:80): the tie tests below apply
                        kw = dict(kw, agc=1, recovery_rate=float(10 ** ((kw["recovery_rate"] * D / FS) / 20)), hangtime=kw["hangtime"] * FS / D)
                    scale = max(float(np.abs(audio).max()), 1e-6)
                    err = float(np.abs(ga - audio).max()) / scale
                    worst = max(worst, err)
                    skip = (2400 // D) * nch if call == 0 and b == 0 else 0
                    st_gain_off = bool(kw.get("agc")) and abs(g_st[c][b]["agc_gain"] - sig["agc_gain"]) > 1e-4 * abs(sig["agc_gain"])
                    pcm_off = int(np.abs(gp.astype(np.int32) - orc.scaleclip(audio).astype(np.int32)).max()) > 1
                    if (err > 2e-4 or st_gain_off or pcm_off) and kw.get("agc") and replay.confirms(c, b, g_aud[c][sl], g_pcm[c][sl], g_st[c][b], skip):
                        ties.append((k, c, float(err), -2, "the reference loop over the GPU's own baseband gives the GPU's output"))
                        continue
                    if err > 2e-4 and kw.get("agc") and (is_agc_tie(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch) or
                                                         (open_tie.get(c) and tie_goes_on(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch))):
                        ties.append((k, c, float(err), -1))  # with a hang time the attacking side holds its gain: many steps
                        open_tie[c] = True  # (a verified tie may reach into the channel's next block)
                        continue
                    if err <= 2e-4:
                        open_tie[c] = False
                    assert err <= 2e-4, (f"ch {c} {kw} block {b} call {call}: audio err {err:.2e}; not an AGC branch tie: " +
                                         (is_agc_tie.why if kw.get("agc") else "no AGC") +
                                         f"; agc_gain gpu {g_st[c][b]['agc_gain']:.6g} oracle {sig['agc_gain']:.6g}")
                    d = np.abs(gp.astype(np.int32) - orc.scaleclip(audio).astype(np.int32))
                    i = int(d.argmax())
                    if d.max() > 1 and kw.get("agc") and (is_agc_tie(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch) or
                                                          (open_tie.get(c) and tie_goes_on(ga, audio, kw["recovery_rate"], kw.get("hangtime", 0.0), nch))):
                        open_tie[c] = True
                        # The hang AGC rides at its threshold (linear.c:121-135: amplitude*gain > headroom -> attack, else
                        # gain *= recovery_rate), so a last-bit difference in the amplitude can flip one sample's branch;
                        # the two gains then differ by one recovery step until the next attack.  Bounded by
                        # recovery_rate - 1 (1.4e-5 at the reference's 6 dB/s: half an LSB at full scale).
                        ties.append((k, c, float(err), int(d.max())))
                        continue  # (the block's status scalars carry the same gain step: not compared for this block)
                    over = np.flatnonzero(d > 1)
                    if over.size and pcm_follows_audio(ga, gp, audio, over):
                        ties.append((k, c, float(err), int(d.max()), "pcm exact for own audio, block over full scale"))
                        continue
                    if d.max() == 2 and all(abs(abs(float(audio[j])) - 1.0) <= 2e-7 and abs(abs(float(ga[j])) - 1.0) <= 2e-7 for j in over):
                        # audio.c:22-28: scaleclip jumps from -32766 (truncation of 32767 x) to -32768 at x = -1, so two
                        # streams one ulp apart AT the clip boundary are 2 LSB apart; the samples are checked to be there
                        ties.append((k, c, float(err), 2, "clip boundary"))
                        continue
                    assert d.max() <= 1, (f"ch {c} {kw} block {b} call {call}: pcm max diff {d.max()} at sample {i}: audio "
                                          f"{ga[i]:.7f} vs {audio[i]:.7f}, block audio err {err:.2e}, neighbours "
                                          f"{np.array2string(ga[max(i - 2, 0):i + 3], precision=6)} vs "
                                          f"{np.array2string(audio[max(i - 2, 0):i + 3], precision=6)}; "
                                          f"{int((d > 1).sum())} samples over")
                    keys = ["bb_power", "agc_gain"] if kw["demod"] == "am" else ["bb_power", "agc_gain", "output_level"]
                    for key in keys:
                        ref = sig[key]
                        if key == "output_level" and call == 0 and b == 0:
                            # the block's first samples are amplified round-off on both sides (skipped above for audio and
                            # PCM) and count in this sum: the status is held against the GPU's OWN audio of the whole
                            # block, in float64, instead (the audio behind the start-up has just been compared)
                            full = np.asarray(g_aud[c][sl], np.float64)
                            ref = float((full * full).sum() / (OLEN * nch))
                            assert abs(g_st[c][b][key] - ref) <= 1e-5 * abs(ref) + 1e-12, (c, kw, key + " vs own audio", b, call, g_st[c][b][key], ref)
                            continue
                        assert abs(g_st[c][b][key] - ref) <= 1e-4 * abs(ref) + 1e-12, (c, kw, key, b, call, g_st[c][b][key], ref)
    finally:
        bank.close()
    return T, calls, worst


def is_agc_tie(ga, audio, recovery_rate, hangtime, nch=1):
    is_agc_tie.why = "no divergence found above the floor"
    return _is_agc_tie(ga, audio, recovery_rate, hangtime, nch)


def _is_agc_tie(ga, audio, recovery_rate, hangtime, nch=1):
    """An independent test of the claim "hang-AGC branch tie" (linear.c:121-135), not just a bound on its size.  When a
    last-bit difference in one sample's amplitude flips attack/recover, the side that recovered is exactly ONE recovery
    step above the side that attacked, at that sample.  From there both follow their own attack/hang/recover sequence
    (the ratio moves by whole steps, or anywhere inside the band when one side attacks alone) until an attack on both
    sides resets both gains to headroom/amplitude.  So: up to the first divergent sample the two audio streams agree; the
    first sample that is off by more than half a step (beyond an absolute floor of 3e-6 of the block's peak: fp32 baseband noise) is off by a whole number of steps, one to three (recovery_rate^k or its inverse, within 20 %); nowhere does it leave recovery_rate^+-(hangtime + 2).  A wrong
    sample or a scale error does not start with exactly one recovery step."""
    ga, audio = np.asarray(ga, np.float64), np.asarray(audio, np.float64)
    if nch == 2:  # stereo: one gain per I/Q pair; take the ratio of the pairs' magnitudes (a lone I or Q may be near zero)
        ga, audio = np.hypot(ga[0::2], ga[1::2]), np.hypot(audio[0::2], audio[1::2])
    big = np.abs(audio) > 1e-3 * max(float(np.abs(audio).max()), 1e-9)
    if big.sum() < 16:
        return False
    ratio = ga[big] / audio[big]
    if (ratio <= 0).any():
        return False
    step = recovery_rate - 1
    # (the streams come from fp32 basebands that differ by ~1e-6 of the block's level: a sample far below the peak carries
    # that as a large RELATIVE difference without any gain step behind it, so the floor is absolute)
    floor = 3e-6 * float(np.abs(audio).max())
    off = np.abs(ga[big] - audio[big]) > 0.5 * step * np.abs(audio[big]) + floor
    if not off.any():
        return False
    i0 = int(np.argmax(off))
    r0 = ratio[i0]
    # whole recovery steps at the first sample that shows: one, or -- when the step itself fell on samples near a zero
    # crossing, below the floor -- the two or three that have accumulated by then
    k_est = abs(np.log(r0)) / np.log(recovery_rate)
    unc = floor / abs(audio[big][i0]) / step  # what the absolute floor leaves undetermined, in steps
    one_step = unc <= 0.6 and k_est - unc > 0.3 and any(abs(k_est - k) <= 0.2 + unc for k in (1, 2, 3))
    if not one_step:
        # the first sample that shows may be a small one (large unc).  What follows the flip is fixed, though: without a
        # hang time both sides recover in step, so the ratio STAYS at its whole number of steps until the next attack;
        # with one, the side that attacked holds its gain while the other goes on recovering, so the ratio climbs by
        # exactly one step per sample for as long as the hang lasts.  Take k from the next few samples that are large
        # enough to pin it (floor below 0.15 of a step); with the ramp removed all of them must give the same k.
        a_big, pos = np.abs(audio[big]), np.flatnonzero(big)
        reach = 60 if hangtime == 0 else int(min(60, hangtime - 3))
        # Which samples can pin it is not taken from the (conservative) floor but shown by the samples themselves: eight
        # of them above 5 % of the peak must agree on k to within half a step peak to peak -- noise of the size the floor
        # allows for would scatter them -- and their median must be a whole number, one to three, to a quarter.
        peak = float(np.abs(audio).max())
        pin = [j for j in range(i0, a_big.size) if pos[j] - pos[i0] <= reach and a_big[j] >= 0.05 * peak][:8]
        if len(pin) >= 8:
            ks = np.abs(np.log(ratio[pin])) / np.log(recovery_rate) - (0 if hangtime == 0 else (pos[pin] - pos[i0]))
            kk = float(np.median(ks))
            if np.ptp(ks) <= 0.5 and any(abs(kk - k) <= 0.25 for k in (1, 2, 3)):
                one_step, k_est, unc = True, kk, float(np.ptp(ks))
    idx = np.flatnonzero(big)[i0]
    is_agc_tie.why = (f"first sample off at {idx}: ratio {r0:.7f} = {k_est:.2f} steps (+-{unc:.2f}), values gpu "
                      f"{np.array2string(ga[max(idx - 2, 0):idx + 4], precision=7)} oracle {np.array2string(audio[max(idx - 2, 0):idx + 4], precision=7)}; "
                      f"ratio range [{ratio.min():.6f}, {ratio.max():.6f}], band {recovery_rate ** (hangtime + 2):.6f}")
    band = recovery_rate ** (hangtime + 2) * (1 + 1e-5)
    return bool(one_step and ratio.max() <= band and ratio.min() >= 1 / band)


class Replay:
    """The reference's demodulator loop (linear.c:57-156 as oracle/ka9q_oracle.c restates it) run over the GPU's OWN
    baseband, channel by channel and block by block from the first: the last word on a block where the GPU and the
    oracle part ways at an AGC knife edge.  If this reproduces the GPU's audio to 2e-6 of the block's peak (the bar
    against the oracle is 2e-4; what is left is the shift oscillator's closed form against its recurrence), its gain
    to 1e-6 and its PCM to the LSB, the only difference between the GPU and the oracle is their two fp32 basebands.
    Mono and envelope channels of the linear demodulator only (a stereo channel's baseband is overwritten by its
    output, linear.c:145)."""

    def __init__(self, chans, out_samprate):
        self.dem = {}
        for c, kw in enumerate(chans):
            if kw["demod"] == "linear" and (kw.get("env") or kw.get("channels", 1) == 1):
                self.dem[c] = orc.LinearDemod(out_samprate, 0, 0, kw.get("agc", 0), kw.get("env", 0), kw.get("channels", 1),
                                              kw["headroom"], kw.get("hangtime", 0.0), kw.get("recovery_rate", 1.0), 5.0,
                                              kw.get("agc_gain", 1.0), kw.get("shift", 0.0))
        self.out = {}

    def run(self, bank, T, olen):
        for c, dem in self.dem.items():
            y = bank.baseband(c, T)
            for b in range(T):
                self.out[(c, b)] = dem.block(y[b * olen:(b + 1) * olen].copy())

    def confirms(self, c, b, ga_full, gp_full, st, skip):
        if (c, b) not in self.out:
            return False
        ra, rs = self.out[(c, b)]
        ga, gp, ra = ga_full[skip:], gp_full[skip:], ra[skip:]
        peak = max(float(np.abs(ra).max()), 1e-9)
        d = np.abs(gp.astype(np.int32) - orc.scaleclip(ra).astype(np.int32))
        return bool(float(np.abs(ga - ra).max()) <= 2e-6 * peak and d.max() <= 1
                    and abs(float(st["agc_gain"]) - rs["agc_gain"]) <= 1e-6 * abs(rs["agc_gain"])
                    and (skip > 0 or abs(float(st["output_level"]) - rs["output_level"]) <= 1e-5 * abs(rs["output_level"]) + 1e-12))


def pcm_follows_audio(ga, gp, audio, over):
    """PCM more than 1 LSB apart in a block whose peak is over full scale (am.c while its DC estimate settles: samples
    of +-300 beside one of 0.7 that is the difference of two such numbers).  The +-1 LSB bar presumes audio within full
    scale; here it is split in two: the GPU's PCM is EXACTLY scaleclip (audio.c:22-28) of the GPU's own float audio,
    every sample, and where the PCM differs the float audio is within fp32 rounding of the block's magnitude
    (2e-6 of its peak), which accounts for the LSBs."""
    peak = float(np.abs(audio).max())
    if peak <= 1.0 or not np.array_equal(orc.scaleclip(np.asarray(ga, np.float32)), gp):
        return False
    dif = np.abs(np.asarray(ga, np.float64) - np.asarray(audio, np.float64))
    d = np.abs(gp.astype(np.int64) - orc.scaleclip(audio).astype(np.int64))
    return bool((dif[over] <= 2e-6 * peak).all() and (d[over] <= 2 + 32767 * dif[over]).all())


def tie_goes_on(ga, audio, recovery_rate, hangtime, nch=1):
    """A block that STARTS inside a divergence whose first sample was verified as a branch tie in the same channel's
    previous block (the caller keeps that record): nothing new can be pinned at its first sample, so the claim here is
    only that the two gains stay inside the band the two attack/hang/recover sequences can span until both attack."""
    ga, audio = np.asarray(ga, np.float64), np.asarray(audio, np.float64)
    if nch == 2:
        ga, audio = np.hypot(ga[0::2], ga[1::2]), np.hypot(audio[0::2], audio[1::2])
    peak = max(float(np.abs(audio).max()), 1e-9)
    big = np.abs(audio) > 1e-2 * peak
    if big.sum() < 16:
        return False
    ratio = ga[big] / audio[big]
    band = recovery_rate ** (hangtime + 2) * (1 + 1e-4)
    return bool((ratio > 0).all() and ratio.max() <= band and ratio.min() >= 1 / band)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global D, OLEN
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    OLEN = L // D
    rng = np.random.default_rng(seed)
    assert hip.lib().ka9q_hip_init(0) == 0, hip.last_error()
    bad = 0
    ties = []
    t0 = time.time()
    for k in range(cases):
        try:
            T, calls, worst = run_case(rng, k, ties)
            print(f"[{k}] ok  T={T} calls={calls} worst audio err {worst:.2e}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"[{k}] BAD {str(e)[:600]}", flush=True)
    for t in ties:
        print("%s: case %d channel %d, block audio err %.2e, pcm diff %d" % ((t[4] if len(t) > 4 else "agc tie",) + tuple(t[:4])))
    print(f"{cases} cases x 6 channels, {bad} bad, {len(ties)} blocks with a verified tie (AGC branch, or the clip boundary of scaleclip), {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
