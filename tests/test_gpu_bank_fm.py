"""GPU parity: the channel bank (kh_bank_*) against the oracle pipeline, FM path (fm.c:43-139).

Bars: baseband <= 1e-5 relative RMS, PCM within +-1 LSB, status scalars within 1e-4 relative."""
import numpy as np
import pytest

import pipeline
import synth
from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _run_pair(hip, L, M, D, fs, chans, x, nblocks_per_call, calls, flat=0, n0=False, referee=None):
    """referee (tools/fuzz_bank.py): called as referee(field, ctx) for every comparison that is outside its bar; True
    means the deviation has been shown to be the reference arithmetic's own (the sweep records it) and the REMAINING
    comparisons still run; None (the fixed tests): any deviation fails at once."""
    C = len(chans)
    T = nblocks_per_call
    bank = hip.Bank(L, M, D, 1, C, T, fs, compute_n0=int(n0), n0_low=-5500.0, n0_high=5500.0)
    pipe = pipeline.Pipeline(L, M, D, fs, compute_n0=n0, n0_low=-5500.0, n0_high=5500.0)
    N = L + M - 1
    rots = []
    for c, (k, *_rest) in enumerate(chans):
        rot = synth.rotate_for(k, 12500.0, N, fs)
        rots.append(rot)
        kw = dict(rotate=rot, low=-5500.0, high=5500.0, kaiser_beta=3.0, flat=flat)
        bank.set_channel(c, demod=hip.KH_DEMOD_FM, **kw)
        pipe.add_channel(demod="fm", **kw)
    olen = L // D
    res = []

    def check(ok, field, msg, **ctx):
        if ok:
            return True
        if referee is not None and referee(field, dict(ctx, L=L, M=M, D=D, fs=fs, N=N, T=T, x=x, rots=rots, bank=bank,
                                                         pipe=pipe, msg=msg)):
            return False
        raise AssertionError(msg)

    for call in range(calls):
        seg = x[call * T * L:(call + 1) * T * L]
        bank.write_input(0, seg)
        bank.execute(T)
        g_bb = [bank.baseband(c, T) for c in range(C)]
        g_pcm = [bank.pcm(c, T) for c in range(C)]
        g_aud = [bank.audio(c, T) for c in range(C)]
        g_st = [bank.status(c, T) for c in range(C)]
        g_ist = bank.input_status(0, T)
        try:
            g_spec0 = bank.spectrum(0, 0)
        except RuntimeError:  # a sweep under KA9Q_HIP_KEEP_SPECTRUM=0: the bank kept only its channels' rows (tools/soak_sparse.sh)
            g_spec0 = None
        for b in range(T):
            outs, istat = pipe.run_block([seg[b * L:(b + 1) * L]])
            if b == 0 and g_spec0 is not None:
                assert rel_rms(g_spec0, pipe.spectrum(0)) < 1e-5
            assert abs(g_ist[b]["if_power"] - istat[0][0]) <= 1e-4 * istat[0][0]
            if n0:
                check(abs(g_ist[b]["n0"] - istat[0][1]) <= 1e-4 * abs(istat[0][1]), "n0",
                      f'["n0"] block {b} call {call}: {g_ist[b]["n0"]} vs {istat[0][1]}', call=call, block=b,
                      gpu=float(g_ist[b]["n0"]), oracle=float(istat[0][1]))
            for c in range(C):
                y, audio, sig = outs[c]
                sl = slice(b * olen, (b + 1) * olen)
                err = rel_rms(g_bb[c][sl], y)
                bb_ok = check(err < 1e-5, "baseband", f"baseband ch {c} block {b} call {call}: {err:.2e}", call=call, block=b,
                              ch=c, gpu=g_bb[c][sl], oracle=y)
                pcm_o = orc_scaleclip(audio)
                d = np.abs(g_pcm[c][sl].astype(np.int32) - pcm_o.astype(np.int32))
                # (a channel whose baseband deviation the referee has accepted is a noise-only channel with the squelch
                # closed: its PCM is zeros on both sides and is still compared)
                if d.max() > 1:
                    i = int(np.argmax(d))
                    e = np.abs(g_aud[c][sl] - audio) - 1e-4 * np.abs(audio)
                    check(False, "fm_pcm", f"pcm ch {c} block {b} call {call}: max diff {d.max()} at sample {i}: audio gpu {g_aud[c][sl][i]!r} "
                          f"oracle {audio[i]!r}", call=call, block=b, ch=c, gpu=g_aud[c][sl], oracle=audio, pcm_gpu=g_pcm[c][sl],
                          y_gpu=g_bb[c][sl], y_oracle=y, flat=flat, worst=i, beyond=int((e > 2e-5).sum()))
                    continue  # (the referee has judged this block's audio; its status scalars follow from it)
                if not np.allclose(g_aud[c][sl], audio, rtol=1e-4, atol=2e-5):
                    e = np.abs(g_aud[c][sl] - audio) - 1e-4 * np.abs(audio)
                    i = int(np.argmax(e))
                    check(False, "fm_audio", f"FM audio ch {c} block {b} call {call}: worst at sample {i} of {olen}: gpu "
                          f"{g_aud[c][sl][max(i - 2, 0):i + 3]} oracle {audio[max(i - 2, 0):i + 3]}; {int((e > 2e-5).sum())} samples "
                          f"beyond the bar; y[0] gpu {g_bb[c][sl][0]!r} oracle {y[0]!r}", call=call, block=b, ch=c, gpu=g_aud[c][sl],
                          oracle=audio, y_gpu=g_bb[c][sl], y_oracle=y, flat=flat, worst=i, beyond=int((e > 2e-5).sum()))
                st = g_st[c][b]
                for key in ("bb_power", "snr", "output_level"):
                    ref = sig[key]
                    # snr: 1e-4 of max(|ref|, 1).  Just above fm_snr's threshold the iteration converges with a
                    # ratio close to 1 and stops on a difference of .001 (fm.c:198), so a last-bit change of
                    # r = avg^2/var (fp32 sums in another order) moves a result of ~0.07 by a few 1e-5; the squelch
                    # compares snr with 4 (fm.c:73)
                    floor = 1e-4 if key == "snr" else 1e-12
                    check(abs(st[key] - ref) <= 1e-4 * abs(ref) + floor, key, f"{key} ch {c} block {b} call {call}: {st[key]} vs {ref}",
                          call=call, block=b, ch=c, gpu=float(st[key]), oracle=float(ref), y_oracle=y, bb_ok=bb_ok, audio_gpu=g_aud[c][sl],
                          audio_oracle=audio)
                res.append((c, b, int(st["squelch_open"]), sig))
                if st["squelch_open"] or sig["snr"] > 4:
                    for key in ("foffset", "pdeviation"):
                        ref = sig[key]
                        assert abs(st[key] - ref) <= 1e-4 * abs(ref) + 2e-3, (key, c, b, st[key], ref)
    bank.close()
    return res


def orc_scaleclip(a):
    import orc
    return orc.scaleclip(a)


def test_fm_bank_8MSps(hip):
    """config 2 shape (8 MS/s, N = 2^18, D = 256 -> 31.25 kS/s), 6 carriers + 12 empty channels, two launches
    of 3 blocks: exercises the four-step FFT, the rotate indexing, the squelch and the state carry."""
    fs, L, M, D = 8_000_000, 131072, 131073, 256
    N = L + M - 1
    sigs = [(-3, 0.02, 450.0, 2500.0), (-2, 0.01, 500.0, 2500.0), (0, 0.05, 300.0, 2500.0), (1, 0.1, 650.0, 2500.0),
            (2, 0.03, 700.0, 1500.0), (4, 0.015, 900.0, 3000.0)]
    T, calls = 3, 2
    x = synth.wideband_fm(0, T * calls * L, fs, N, sigs)
    # slots 9 .. 20: noise only, squelch stays closed.  Their amplitude statistics sit just above fm_snr's threshold
    # r = pi/(4-pi), where its iteration runs longest (72 such blocks here; the status bar of 1e-4 applies to snr)
    chans = sigs + [(k, 0, 0, 0) for k in range(9, 21)]
    res = _run_pair(hip, L, M, D, fs, chans, x, T, calls, n0=True)
    closed = [r for r in res if r[0] >= 6]
    assert len(closed) == 12 * T * calls and all(r[2] == 0 for r in closed)
    opened = [r for r in res if r[0] == 3 and r[1] >= 2]
    assert all(abs(r[3]["pdeviation"] - 2500) < 150 for r in opened)


def test_fm_squelch_from_r_alone_equals_the_iteration(hip, monkeypatch):
    """kh_fm_demod2 decides a block's squelch (fm.c:73-78: is fm_snr above 4) from r = avg^2/variance alone when r is
    outside [6.15, 6.20] -- fm_snr crosses 4 at r = 6.174478 -- and runs fm_snr's iteration before deciding only inside it;
    the SNR itself is iterated by other wavefronts and only reported.  With KA9Q_HIP_FM2_ITERATE_ALL=1 every block's decision
    comes from the iteration: PCM, audio and status of both runs must be identical bit for bit, over carriers whose SNR
    ranges from far below to far above the threshold, and both must meet the oracle."""
    fs, L, M, D = 8_000_000, 131072, 131073, 256
    N = L + M - 1
    # amplitudes around the squelch threshold (noise sigma 0.01 per component: 4 = 6 dB in a 11 kHz channel)
    sigs = [(k - 8, a, 300.0 + 40 * k, 2500.0) for k, a in enumerate([0.0004, 0.0006, 0.0008, 0.001, 0.0012, 0.0014, 0.0016, 0.0018,
                                                                       0.002, 0.0023, 0.0026, 0.003, 0.004, 0.006, 0.01, 0.05])]
    T, calls = 4, 2
    x = synth.wideband_fm(0, T * calls * L, fs, N, sigs)
    runs = {}
    for name, env in (("r", "0"), ("iterate", "1")):
        monkeypatch.setenv("KA9Q_HIP_FM2_ITERATE_ALL", env)  # read at every launch
        bank = hip.Bank(L, M, D, 1, len(sigs), T, fs)
        for c, (k, _, _, _) in enumerate(sigs):
            bank.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0, kaiser_beta=3.0)
        out = []
        for call in range(calls):
            bank.write_input(0, x[call * T * L:(call + 1) * T * L])
            bank.execute(T)
            out.append((np.stack([bank.pcm(c, T) for c in range(len(sigs))]),
                        [bank.status(c, T) for c in range(len(sigs))]))
        bank.close()
        runs[name] = out
    opened = 0
    for (pa, sa), (pb, sb) in zip(runs["r"], runs["iterate"]):
        assert np.array_equal(pa, pb)
        for a, b in zip(sa, sb):
            for f in ("snr", "squelch_open", "bb_power", "output_level"):
                assert np.array_equal(np.asarray(a[f]), np.asarray(b[f])), f
            opened += int(np.asarray(a["squelch_open"]).sum())
    assert 0 < opened < len(sigs) * T * calls  # both sides of the threshold occur
    _run_pair(hip, L, M, D, fs, sigs, x, T, calls)


@pytest.mark.parametrize("shape", [(8_000_000, 131072, 131073, 256, 5), (1_000_000, 8192, 8193, 32, 7)])
def test_fm_demodulator_arrangements_are_bit_identical(hip, monkeypatch, shape):
    """kh_fm_demod2 hands a workgroup 1 .. 8 channels (8 in banks of >= 1024 FM channels) and 1 .. 16 wavefronts, which
    split into SNR and work wavefronts and take blocks by ticket.  None of that may show in the results: the same bank
    run under forced arrangements (KA9Q_HIP_FM2_CPW x KA9Q_HIP_FM2_WAVES, including one wavefront that does everything,
    channel counts that leave a workgroup partly empty, and more wavefronts than blocks) gives the same PCM, status and
    carried state bit for bit over two launches; the default arrangement is held to the oracle by the tests around this
    one.  13 channels: busy ones (squelch open throughout, opening late, closing), idle ones, a flat one."""
    fs, L, M, D, T = shape
    N = L + M - 1
    calls = 2
    nslots = int(fs / 2 / 12500.0) - 1
    ks = [k for k in (-5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7) if abs(k) <= nslots]
    amps = {-5: 0.05, -3: 0.0021, -2: 0.0024, 0: 0.1, 2: 0.0018, 3: 0.02, 6: 0.003}
    sigs = [(k, amps[k], 300.0 + 35 * (k + 5), 2500.0) for k in ks if k in amps]
    x = synth.wideband_fm(0, T * calls * L, fs, N, sigs)
    x[: T * L // 2] *= 0.02  # the weaker carriers' squelches open during the first launch
    results = {}
    # (third number: KA9Q_HIP_FM2_SPLIT, the workgroups a channel's blocks are dealt over -- what small banks do by themselves)
    for cpw, waves, split in ((0, 0, 1), (1, 1, 1), (1, 2, 1), (1, 16, 1), (2, 1, 1), (2, 3, 1), (4, 4, 1), (8, 1, 1), (8, 16, 1),
                              (3, 5, 1), (0, 0, 0), (0, 0, 2), (1, 1, 3), (1, 4, 5), (1, 16, 7), (1, 2, 4)):
        monkeypatch.setenv("KA9Q_HIP_FM2_CPW", str(cpw))
        monkeypatch.setenv("KA9Q_HIP_FM2_WAVES", str(waves))
        monkeypatch.setenv("KA9Q_HIP_FM2_SPLIT", str(split))
        bank = hip.Bank(L, M, D, 1, len(ks), T, fs)
        for c, k in enumerate(ks):
            bank.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0,
                             kaiser_beta=3.0, flat=int(k == 3))
        out = []
        for call in range(calls):
            bank.write_input(0, x[call * T * L:(call + 1) * T * L])
            bank.execute(T)
            out.append(np.stack([bank.pcm(c, T) for c in range(len(ks))]))
            for c in range(len(ks)):
                st = bank.status(c, T)
                out.append(np.concatenate([np.asarray(st[f], np.float64).ravel() for f in
                                           ("bb_power", "snr", "foffset", "pdeviation", "output_level", "squelch_open")]))
        bank.close()
        results[(cpw, waves, split)] = out
    ref = results[(0, 0, 1)]
    assert np.abs(ref[0]).max() > 500 and any(np.abs(r).max() == 0 for r in ref[0])  # open and squelched channels
    for key, out in results.items():
        for a, b in zip(ref, out):
            assert np.array_equal(a, b, equal_nan=True), key


@pytest.mark.parametrize("shape", [(8_000_000, 131072, 131073, 256, 32, 1), (1_000_000, 8192, 8193, 32, 16, 3)])
def test_fm_demodulator_split_channels_are_bit_identical(hip, monkeypatch, shape):
    """A small bank deals a channel's blocks over neighbouring workgroups of kh_fm_demod2 (a run of blocks each; the runs'
    integrator carry-outs cross through global memory: fm.c:110-111 is linear in lastaudio).  BASELINE configs[1]'s launch
    shape (32 blocks of 512 samples; 4 workgroups per channel by default) and 16 blocks of 256: PCM, status and the state
    carried into the next launch must not depend on the split -- none (1), the default (0), 2, 8, and one that does not
    divide the block count.  Carriers that stay open through every run, one that opens in the middle of a launch (its
    integrator then starts inside a later run), one that closes (decaying carry across runs), idle channels."""
    fs, L, M, D, T, _ = shape
    N = L + M - 1
    ks = [-3, -2, -1, 0, 1, 2, 3]
    steady = [(k, a, 300.0 + 35 * (k + 5), 2500.0) for k, a in ((0, 0.1), (1, 0.02), (3, 0.004))]
    keyed = [(k, a, 300.0 + 35 * (k + 5), 2500.0) for k, a in ((-3, 0.05), (-2, 0.01))]
    n = 2 * T * L
    gate = np.ones(n, np.float32)
    gate[: 5 * T * L // 8] = 0                      # the keyed carriers come up in the middle of the first launch
    gate[11 * T * L // 8: 13 * T * L // 8] = 0      # and drop for a while in the second: a decaying carry across the runs
    x = (synth.wideband_fm(0, n, fs, N, steady) + gate * synth.wideband_fm(0, n, fs, N, keyed, sigma=0.0)).astype(np.complex64)
    results = {}
    for split in (1, 0, 2, 8, 5):
        monkeypatch.setenv("KA9Q_HIP_FM2_SPLIT", str(split))
        bank = hip.Bank(L, M, D, 1, len(ks), T, fs)
        for c, k in enumerate(ks):
            bank.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0,
                             kaiser_beta=3.0, flat=int(k == 1))
        out = []
        for call in range(2):
            bank.write_input(0, x[call * T * L:(call + 1) * T * L])
            bank.execute(T)
            out.append(np.stack([bank.pcm(c, T) for c in range(len(ks))]))
            for c in range(len(ks)):
                st = bank.status(c, T)
                out.append(np.concatenate([np.asarray(st[f], np.float64).ravel() for f in
                                           ("bb_power", "snr", "foffset", "pdeviation", "output_level", "squelch_open")]))
        bank.close()
        results[split] = out
    ref = results[1]
    assert np.abs(ref[0]).max() > 500 and any(np.abs(r).max() == 0 for r in ref[0])  # open and squelched channels
    sq = ref[1][-T:]  # channel 0's (a keyed carrier's) squelch_open over the first launch: closed, then open
    assert sq[0] == 0 and sq[-1] == 1
    sq2 = ref[1 + len(ks) + 1][-T:]  # and over the second: open, closed, open again
    assert sq2[0] == 1 and sq2.min() == 0 and sq2[-1] == 1
    for key, out in results.items():
        for a, b in zip(ref, out):
            assert np.array_equal(a, b, equal_nan=True), key


def test_fm_bank_ndec512_odd_block_counts(hip):
    """N_dec = 512 / olen = 256 (config 4's channel shape at a small N): two blocks of a channel share a wavefront in the
    channel kernel, so odd block counts leave a padding half; compute_n0 rides in the same launch."""
    fs, L, M, D = 1_000_000, 8192, 8193, 32
    N = L + M - 1
    sigs = [(-2, 0.05, 400.0, 2500.0), (0, 0.02, 700.0, 2000.0), (3, 0.1, 1000.0, 3000.0)]
    for T, calls in ((3, 2), (1, 3), (4, 1)):
        x = synth.wideband_fm(0, T * calls * L, fs, N, sigs)
        _run_pair(hip, L, M, D, fs, sigs + [(7, 0, 0, 0)], x, T, calls, n0=True)


def test_fm_bank_192k_flat_and_deemph(hip):
    """config 1 shape (192 kS/s, L=3840, M=4353, N=8192, D=4): single-workgroup FFT path, olen = 960."""
    fs, L, M, D = 192_000, 3840, 4353, 4
    N = L + M - 1
    T, calls = 5, 2
    x = synth.noise(T * calls * L, 0.01, 3) + synth.fm_carrier(0, T * calls * L, fs, 0.0, 0.5, 1000.0, 3000.0)
    for flat in (0, 1):
        bank_chans = [(0, 0, 0, 0)]
        # +-8 kHz filter of the reference's FM preset (main.c:132-146) would need other edges; the helper
        # uses +-5.5 kHz, still wide enough for 3 kHz deviation + 1 kHz tone by Carson's rule.
        _run_pair(hip, L, M, D, fs, bank_chans, x.astype(np.complex64), T, calls, flat=flat)


def test_fm_bank_reference_default_sizes(hip):
    """What `radio` computes by default at 192 kS/s today: L = 3840, N = nextfastfft(2L-1) = 7680 = 2^9*3*5, M = 3841,
    D = 4 -> N_dec = 1920 (main.c:326-330, misc.c:129-148): the mixed-radix transforms (kh_anyfft.cuh) under the whole FM
    chain, with compute_n0, over two launches."""
    fs, L, M, D = 192_000, 3840, 3841, 4
    T, calls = 4, 2
    x = synth.noise(T * calls * L, 0.01, 3) + synth.fm_carrier(0, T * calls * L, fs, 0.0, 0.5, 1000.0, 3000.0)
    for flat in (0, 1):
        _run_pair(hip, L, M, D, fs, [(0, 0, 0, 0)], x.astype(np.complex64), T, calls, flat=flat, n0=True)


def test_fm_squelch_transitions_and_no_baseband_mode(hip):
    """A carrier keyed on and off block by block: the squelch opens, holds its one-block tail and closes (fm.c:73-78),
    `state` is zeroed while closed (fm.c:123), so sample 0 of a block is demodulated against the last sample of the
    block before, against zero, or against the state carried from the launch before.  Every variant must match the
    oracle, in both de-emphasis modes; and with kh_bank_keep_baseband(0) the PCM and status must not change by a bit."""
    fs, L, M, D = 192_000, 3840, 4353, 4
    olen = L // D
    T, calls = 6, 3
    n = T * calls * L
    keyed = np.array([1, 1, 0, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 0, 0, 0, 1, 1], np.float32)  # per block
    env = np.repeat(keyed, L)
    x = (synth.noise(n, 0.01, 5) + env * synth.fm_carrier(0, n, fs, 0.0, 0.4, 700.0, 2500.0)).astype(np.complex64)
    for flat in (0, 1):
        kw = dict(rotate=0, low=-5500.0, high=5500.0, kaiser_beta=3.0, flat=flat)
        bank = hip.Bank(L, M, D, 1, 1, T, fs)
        lean = hip.Bank(L, M, D, 1, 1, T, fs)
        lean.keep_baseband(0)
        pipe = pipeline.Pipeline(L, M, D, fs)
        bank.set_channel(0, demod=hip.KH_DEMOD_FM, **kw)
        lean.set_channel(0, demod=hip.KH_DEMOD_FM, **kw)
        pipe.add_channel(demod="fm", **kw)
        seen = set()
        for call in range(calls):
            seg = x[call * T * L:(call + 1) * T * L]
            for bk in (bank, lean):
                bk.write_input(0, seg)
                bk.execute(T)
            pcm, st = bank.pcm(0, T), bank.status(0, T)
            assert np.array_equal(pcm, lean.pcm(0, T)) and st.tobytes() == lean.status(0, T).tobytes()
            for b in range(T):
                outs, _ = pipe.run_block([seg[b * L:(b + 1) * L]])
                _y, audio, sig = outs[0]
                sl = slice(b * olen, (b + 1) * olen)
                d = np.abs(pcm[sl].astype(np.int32) - orc_scaleclip(audio).astype(np.int32))
                if flat:
                    # without the amplitude weight of fm.c:91-93 the angle of a near-zero sample pair is ill-conditioned
                    # (a 1e-7 baseband difference moves arg() of a 1e-4 sample by 1e-3 rad): the +-1 LSB bar is applied
                    # where both samples carry at least 2 % of the block's RMS amplitude, 16 LSB elsewhere
                    a = np.abs(_y)
                    ok = np.minimum(a, np.concatenate(([a[0]], a[:-1]))) > 0.02 * np.sqrt(np.mean(a * a))
                    assert d[ok].max() <= 1 and d.max() <= 16, (flat, call, b, int(d[ok].max()), int(d.max()))
                else:
                    assert d.max() <= 1, (flat, call, b, int(d.max()))
                so = int(pipe.channels[0].dem.st.squelch_open)  # the oracle's counter after this block
                assert int(st[b]["squelch_open"]) == so, (flat, call, b)
                seen.add(so)
                if sig["snr"] > 4:
                    for key in ("foffset", "pdeviation"):
                        assert abs(st[b][key] - sig[key]) <= 1e-4 * abs(sig[key]) + 2e-3, (key, flat, call, b)
        assert seen == {0, 1}
        bank.close()
        lean.close()


def test_rtp_packetisation_plan(hip):
    """audio.c:32-135 on the device: a keyed FM carrier gives blocks of silence (closed squelch,
    the de-emphasis tail decays below one LSB), a stereo linear channel gives 4 chunks of 240 frames per 960-sample block.
    The plan must equal the packetiser restatement run over the bank's own PCM, chunk for chunk, across two launches
    (timestamp, sequence number and the silent flag carry over), starting from a non-trivial RTP state."""
    import orc
    fs, L, M, D = 192_000, 3840, 4353, 4
    olen = L // D
    T, calls = 6, 2
    n = T * calls * L
    keyed = np.array([1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1], np.float32)
    x = (synth.noise(n, 0.0, 5) + np.repeat(keyed, L) * synth.fm_carrier(0, n, fs, 0.0, 0.4, 700.0, 2500.0)).astype(np.complex64)
    bank = hip.Bank(L, M, D, 1, 3, T, fs)
    bank.rtp_enable(1)
    bank.set_channel(0, demod=hip.KH_DEMOD_FM, rotate=0, low=-5500.0, high=5500.0, kaiser_beta=3.0)
    bank.set_channel(1, demod=hip.KH_DEMOD_LINEAR, rotate=0, low=-5000.0, high=5000.0, kaiser_beta=3.0, channels=2, agc=0,
                     agc_gain=1.0)
    bank.set_channel(2, rotate=0, low=-5000.0, high=5000.0)  # no demodulator: no PCM, no chunks
    bank.rtp_set_state(0, timestamp=0xFFFFF000, seq=0xFFFE, silent=1)  # both counters wrap inside the run
    ref = [orc.RtpPacketiser(1, 0xFFFFF000, 0xFFFE, 1), orc.RtpPacketiser(2)]
    sent = [0, 0]
    for call in range(calls):
        bank.write_input(0, x[call * T * L:(call + 1) * T * L])
        bank.execute(T)
        for ch, chans in ((0, 1), (1, 2)):
            pcm = bank.pcm(ch, T)  # (2 words per output sample for the stereo channel)
            plan, after = bank.rtp(ch)
            want = []
            wpb = olen * chans
            for b in range(T):
                want += ref[ch].block(pcm[b * wpb:(b + 1) * wpb], base_offset=b * wpb)
            assert len(plan) == len(want) == T * ((wpb + 479) // 480)
            for got, w in zip(plan, want):
                ts, seq, marker, send, off, nw = w
                assert (int(got["timestamp"]), int(got["send"]), int(got["offset"]), int(got["nwords"])) == (ts, send, off, nw)
                if send:
                    assert (int(got["seq"]), int(got["marker"])) == (seq, marker)
            r = ref[ch]
            assert (int(after["timestamp"]), int(after["seq"]), int(after["silent"]), int(after["packets"]), int(after["bytes"]),
                    int(after["samples"])) == (r.timestamp, r.seq, r.silent, r.packets, r.bytes, r.samples)
            sent[ch] += int(plan["send"].sum())
        assert len(bank.rtp(2)[0]) == 0
    # the keyed carrier produced both sent and suppressed chunks, and at least one marker after silence
    assert 0 < sent[0] < calls * T * 2
    bank.close()
