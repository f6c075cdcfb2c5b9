"""GPU parity on the BASELINE.json configurations 3, 4 and 5 (SURVEY.md 8 shape table).

The oracle is run on a handful of channels per configuration (it needs ~0.1 s per 2^20-point block); the full
channel counts are covered through size-independent properties: channels are independent (a channel's output does
not depend on which other channels exist), identical channels give identical bits, and spot-checked channels of
the full-size bank match the oracle."""
import numpy as np
import pytest

import orc
import pipeline
import synth
from conftest import rel_rms

pytestmark = pytest.mark.gpu

HEADROOM = 10 ** (-15 / 20)
AGC0 = 10 ** (80 / 20)


def _recovery(out_rate):
    return 10 ** ((6.0 / out_rate) / 20)  # +6 dB/s (radio.c:398)


def _check_channel(name, kind, g_bb, g_pcm, g_st, outs_by_block, olen, nch=1, settle=0, skip0=0):
    for b, (y, audio, sig) in enumerate(outs_by_block):
        sl = slice(b * olen, (b + 1) * olen)
        if kind != "linear_pll":  # linear.c:91 rewrites the baseband when the PLL runs
            if not (kind.startswith("linear") and nch == 2):  # linear.c:145 rewrites it for stereo output
                assert rel_rms(g_bb[sl], y) < 1e-5, (name, b)
        if b < settle:
            continue
        slp = slice(b * olen * nch, (b + 1) * olen * nch)
        gp, want = g_pcm[slp], orc.scaleclip(audio)
        if b == 0 and skip0:
            gp, want = gp[skip0 * nch:], want[skip0 * nch:]
        d = np.abs(gp.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1, (name, b, int(d.max()))
        ref = sig["bb_power"]
        assert abs(g_st[b]["bb_power"] - ref) <= 1e-4 * abs(ref) + 1e-15, (name, b)


def _n0_float64(spec, N, low, high, samprate):
    """compute_n0 (radio.c:422-465) with the two means accumulated in float64 instead of serially in float32."""
    low_n = int(np.float32(N * np.float32(low)) / np.float32(samprate))
    high_n = int(np.float32(N * np.float32(high)) / np.float32(samprate))
    bincount = N - (high_n - low_n)
    idx = (high_n + 1 + np.arange(bincount)) % N
    p = (spec[idx].real.astype(np.float32) ** 2 + spec[idx].imag.astype(np.float32) ** 2).astype(np.float64)
    avg = p.mean()
    sel = p < 2 * avg
    return p[sel].mean() / (float(N) * samprate)


def test_config3_20MSps_mixed_modes(hip):
    """20 MS/s, N = 2^19, D = 512 -> 39.0625 kS/s; k mod 4 = 0,1 NBFM, 2 envelope AM with AGC, 3 USB with hang AGC."""
    fs, L, M, D = 20_000_000, 262144, 262145, 512
    N = L + M - 1
    olen = L // D
    out_rate = fs / D
    T, calls = 2, 2
    n = T * calls * L
    ks = [-6, -5, -2, -1, 0, 1, 2, 3]
    x = synth.noise(n, 0.01, 31)
    cfgs = []
    for k in ks:
        rot = synth.rotate_for(k, 12500.0, N, fs)
        fc = rot * fs / N
        m = k % 4
        if m in (0, 1):
            x = x + synth.fm_carrier(0, n, fs, fc, 0.05, 300.0 + 50 * (k % 8), 2500.0, phase0=0.1 * k)
            cfgs.append(("fm", dict(demod="fm", rotate=rot, low=-5500.0, high=5500.0, headroom=HEADROOM)))
        elif m == 2:
            x = x + synth.am_carrier(0, n, fs, fc, 0.05, 400.0, 0.5, phase0=0.3)
            cfgs.append(("linear_env", dict(demod="linear", rotate=rot, low=-5000.0, high=5000.0, env=1, agc=1,
                                            hangtime=1.1 * out_rate, recovery_rate=_recovery(out_rate), agc_gain=AGC0,
                                            headroom=HEADROOM)))
        else:
            x = x + synth.usb_two_tone(0, n, fs, fc, 0.05)
            cfgs.append(("linear_usb", dict(demod="linear", rotate=rot, low=100.0, high=3000.0, agc=1,
                                            hangtime=1.1 * out_rate, recovery_rate=_recovery(out_rate), agc_gain=AGC0,
                                            headroom=HEADROOM)))
    x = x.astype(np.complex64)
    C = len(cfgs)
    bank = hip.Bank(L, M, D, 1, C, T, fs)
    pipe = pipeline.Pipeline(L, M, D, fs)
    did = {"fm": hip.KH_DEMOD_FM, "linear": hip.KH_DEMOD_LINEAR}
    for c, (_kind, kw) in enumerate(cfgs):
        kwb = dict(kw)
        kwb["demod"] = did[kw["demod"]]
        bank.set_channel(c, **kwb)
        pipe.add_channel(**kw)
    for call in range(calls):
        seg = x[call * T * L:(call + 1) * T * L]
        bank.write_input(0, seg)
        bank.execute(T)
        per_block = [pipe.run_block([seg[b * L:(b + 1) * L]])[0] for b in range(T)]
        for c, (kind, _kw) in enumerate(cfgs):
            _check_channel(f"c3 ch{c} {kind} call{call}", kind, bank.baseband(c, T), bank.pcm(c, T), bank.status(c, T),
                           [pb[c] for pb in per_block], olen, skip0=300 if call == 0 else 0)
    bank.close()


def test_config4_64MSps_2pow20(hip):
    """64 MS/s, N = 2^20, D = 2048 -> 31.25 kS/s, N_dec = 512, olen = 256; NBFM channels across the whole band
    including both band edges (rotate near +-N/2 exercises the modular wrap of filter.c:278-289)."""
    fs, L, M, D = 64_000_000, 524288, 524289, 2048
    N = L + M - 1
    olen = L // D
    T, calls = 2, 2
    n = T * calls * L
    ks = [-2559, -2048, -1, 0, 1, 777, 2047, 2559]
    sigs = [(k, 0.03 + 0.005 * i, 300.0 + 70 * i, 2500.0) for i, k in enumerate(ks)]
    x = synth.wideband_fm(0, n, fs, N, sigs)
    C = len(ks)
    bank = hip.Bank(L, M, D, 1, C, T, fs, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
    pipe = pipeline.Pipeline(L, M, D, fs, compute_n0=True, n0_low=-5500.0, n0_high=5500.0)
    for c, k in enumerate(ks):
        kw = dict(rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0, headroom=HEADROOM)
        bank.set_channel(c, demod=hip.KH_DEMOD_FM, **kw)
        pipe.add_channel(demod="fm", **kw)
    n0_prev = None
    for call in range(calls):
        seg = x[call * T * L:(call + 1) * T * L]
        bank.write_input(0, seg)
        bank.execute(T)
        ist = bank.input_status(0, T)
        per_block = []
        for b in range(T):
            outs, istat = pipe.run_block([seg[b * L:(b + 1) * L]])
            per_block.append(outs)
            # radio.c:450-462 adds ~10^6 float32 terms one after the other: at this size the serial sum itself is
            # only good to ~sqrt(n)*eps ~ 1e-4 (and the reference's -funsafe-math build re-associates it anyway), so
            # the oracle is held to 1e-3 and the 1e-4 bar is applied against the same definition summed in float64
            assert abs(ist[b]["n0"] - istat[0][1]) <= 1e-3 * abs(istat[0][1]), (call, b)
            n0_64 = _n0_float64(pipe.spectrum(0), N, -5500.0, 5500.0, fs)
            n0_prev = n0_64 if n0_prev is None else n0_prev + .005 * (n0_64 - n0_prev)  # radio.c:199-202
            assert abs(ist[b]["n0"] - n0_prev) <= 1e-4 * n0_prev, (call, b, float(ist[b]["n0"]), n0_prev)
            if b == 0:
                assert rel_rms(bank.spectrum(0, 0), pipe.spectrum(0)) < 1e-5
        for c in range(C):
            _check_channel(f"c4 ch{c} call{call}", "fm", bank.baseband(c, T), bank.pcm(c, T), bank.status(c, T),
                           [pb[c] for pb in per_block], olen)
    bank.close()


@pytest.mark.parametrize("C", [512, 1536])
def test_config4_full_size_properties(hip, C):
    """512 channels at N = 2^20 (one GPU's share of config 4), 4 blocks: (i) channel results do not depend on the
    other channels: a spot-checked channel equals the same channel computed alone; (ii) duplicate channels are
    bit-identical; (iii) three spot-checked channels match the oracle.  1536 channels: the same with the FM demodulator's
    arrangement for large banks (eight channels per workgroup from 1024 FM channels on)."""
    fs, L, M, D = 64_000_000, 524288, 524289, 2048
    N = L + M - 1
    olen = L // D
    T = 4
    ks = list(range(-C // 2, C // 2))
    ks[500] = ks[10]  # a duplicate
    strong = {ks[10]: 0.05, ks[100]: 0.03, ks[333]: 0.04}
    sigs = [(k, a, 400.0 + (k % 7) * 50, 2500.0) for k, a in strong.items()]
    x = synth.wideband_fm(0, T * L, fs, N, sigs)
    bank = hip.Bank(L, M, D, 1, C, T, fs)
    for c, k in enumerate(ks):
        bank.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0)
    bank.write_input(0, x)
    bank.execute(T)
    full = {c: (bank.baseband(c, T), bank.pcm(c, T)) for c in (10, 100, 333, 500, 0, C - 1)}
    bank.close()
    assert np.array_equal(full[10][0], full[500][0]) and np.array_equal(full[10][1], full[500][1])
    assert np.abs(full[10][1]).max() > 500 and np.abs(full[0][1]).max() == 0  # open / squelched
    solo = hip.Bank(L, M, D, 1, 1, T, fs)
    solo.set_channel(0, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(ks[333], 12500.0, N, fs), low=-5500.0, high=5500.0)
    solo.write_input(0, x)
    solo.execute(T)
    assert np.array_equal(solo.baseband(0, T), full[333][0]) and np.array_equal(solo.pcm(0, T), full[333][1])
    solo.close()
    pipe = pipeline.Pipeline(L, M, D, fs)
    for c in (10, 100, 333):
        pipe.add_channel(demod="fm", rotate=synth.rotate_for(ks[c], 12500.0, N, fs), low=-5500.0, high=5500.0)
    for b in range(T):
        outs, _ = pipe.run_block([x[b * L:(b + 1) * L]])
        for j, c in enumerate((10, 100, 333)):
            y, audio, _sig = outs[j]
            sl = slice(b * olen, (b + 1) * olen)
            assert rel_rms(full[c][0][sl], y) < 1e-5, (c, b)
            assert np.abs(full[c][1][sl].astype(np.int32) - orc.scaleclip(audio).astype(np.int32)).max() <= 1, (c, b)


def _c5_case(i, fs):
    rng = np.random.default_rng(1000 + i)
    f0 = float(rng.uniform(-10e3, 10e3))
    fdot = float(rng.uniform(-200.0, 200.0))
    return f0, fdot, bool(i % 2)  # odd pipelines: suppressed carrier, squaring loop


def _c5_signal(i, n, fs):
    f0, fdot, square = _c5_case(i, fs)
    nz = synth.noise(n, 0.001, 200 + i)
    if square:
        t = np.arange(n, dtype=np.float64) / fs
        ph = 2 * np.pi * ((48_000.0 + f0) * t + 0.5 * fdot * t * t) + 0.4
        return (nz + (0.1 * np.cos(2 * np.pi * 500 * t) * np.exp(1j * ph)).astype(np.complex64)).astype(np.complex64)
    return (nz + synth.swept_carrier(0, n, fs, 48_000.0 + f0, fdot, 0.1)).astype(np.complex64)


def _c5_channel_kw(i):
    _f0, _fdot, square = _c5_case(i, 192000)
    return dict(demod="linear", input=i, low=-5000.0, high=5000.0, pll=1, square=int(square), agc=1,
                hangtime=1.1 * 48000, recovery_rate=_recovery(48000), agc_gain=AGC0, headroom=HEADROOM)


def test_config5_pipelines_doppler_pll_agc(hip):
    """Independent 192 kS/s pipelines in the reference's order: Doppler NCO with sweep (set_doppler, radio.c:233-240,
    sign convention of radio.c:238) -> LO2 NCO -> filter -> demod_linear with PLL + AGC; state carried over 5 launches
    of 5 blocks (0.5 s: NCO renormalisation every 16384 samples, PLL pull-in, lock counter running)."""
    fs, L, M, D = 192_000, 3840, 4353, 4
    olen = L // D
    P, T, calls = 6, 5, 5
    n = T * calls * L
    sig = [_c5_signal(i, n, fs) for i in range(P)]
    bank = hip.Bank(L, M, D, P, P, T, fs)
    pipe = pipeline.Pipeline(L, M, D, fs, n_inputs=P)
    for i in range(P):
        f0, fdot, _sq = _c5_case(i, fs)
        kw = _c5_channel_kw(i)
        kwb = dict(kw)
        kwb["demod"] = hip.KH_DEMOD_LINEAR
        bank.set_channel(i, **kwb)
        pipe.add_channel(**kw)
        # open-loop correction: the negated Doppler (radio.c:238) and the -48 kHz second LO
        args = (-48_000.0 / fs, -f0 / fs, -(fdot / fs) / fs)
        bank.set_lo(i, *args)
        pipe.set_lo(i, *args)
    SETTLE = 10  # blocks: PLL start-up on amplified round-off, see tests/test_gpu_bank_linear.py
    for call in range(calls):
        for i in range(P):
            bank.write_input(i, sig[i][call * T * L:(call + 1) * T * L])
        bank.execute(T)
        g_aud = [bank.audio(i, T) for i in range(P)]
        g_pcm = [bank.pcm(i, T) for i in range(P)]
        g_st = [bank.status(i, T) for i in range(P)]
        for b in range(T):
            lo = (call * T + b) * L
            outs, _ = pipe.run_block([sig[i][lo:lo + L] for i in range(P)])
            blk = call * T + b
            for i in range(P):
                _y, audio, s = outs[i]
                sl = slice(b * olen, (b + 1) * olen)
                if blk == 0:
                    continue
                scale = max(float(np.abs(audio).max()), 1e-6)
                err = float(np.abs(g_aud[i][sl] - audio).max()) / scale
                assert err <= (2e-2 if blk < SETTLE else 3e-4), (i, blk, err)
                if blk < SETTLE:
                    continue
                assert np.abs(g_pcm[i][sl].astype(np.int32) - orc.scaleclip(audio).astype(np.int32)).max() <= 1, (i, blk)
                st = g_st[i][b]
                assert st["pll_lock"] == s["pll_lock"] and st["lock_timer"] == s["lock_timer"]
                assert abs(st["agc_gain"] - s["agc_gain"]) <= 1e-4 * abs(s["agc_gain"])
                assert abs(st["foffset"] - s["foffset"]) <= 1e-3 * abs(s["foffset"]) + 2e-3, (i, blk, st["foffset"], s["foffset"])
    # the open-loop Doppler correction left a residual the PLL can hold: |foffset| is a few Hz, not kHz
    assert all(abs(g_st[i][T - 1]["foffset"]) < 50.0 for i in range(P))
    bank.close()


def test_config5_full_256_pipelines_independent(hip):
    """256 pipelines in one bank: pipelines 0..5 of the full bank produce the same bits as the 6-pipeline bank of the
    previous test's first launch (per-lane recurrences must not leak between lanes or wavefronts)."""
    fs, L, M, D = 192_000, 3840, 4353, 4
    P, T = 256, 5
    n = T * L
    small = hip.Bank(L, M, D, 6, 6, T, fs)
    big = hip.Bank(L, M, D, P, P, T, fs)
    sigs = {}
    for i in range(P):
        j = i % 6 if i >= 6 else i  # inputs repeat with period 6 above pipeline 5: identical channels, identical bits
        if j not in sigs:
            sigs[j] = _c5_signal(j, n, fs)
        f0, fdot, _sq = _c5_case(j, fs)
        kw = _c5_channel_kw(j)
        kw["demod"] = hip.KH_DEMOD_LINEAR
        kw["input"] = i
        big.set_channel(i, **kw)
        big.set_lo(i, -48_000.0 / fs, -f0 / fs, -(fdot / fs) / fs)
        big.write_input(i, sigs[j])
        if i < 6:
            small.set_channel(i, **kw)
            small.set_lo(i, -48_000.0 / fs, -f0 / fs, -(fdot / fs) / fs)
            small.write_input(i, sigs[j])
    small.execute(T)
    big.execute(T)
    for i in range(6):
        assert np.array_equal(small.pcm(i, T), big.pcm(i, T)), i
        assert np.array_equal(big.pcm(i, T), big.pcm(i + 6 * 41, T)), i  # pipeline 246+i repeats pipeline i
        assert small.status(i, T).tobytes() == big.status(i, T).tobytes()
    small.close()
    big.close()


@pytest.mark.parametrize("pipeline_env", ["1", "0"])
def test_back_to_back_launches_match_synchronised_ones(hip, monkeypatch, pipeline_env):
    """(pipeline_env "0": KA9Q_HIP_PIPELINE=0, both halves and the raw-ring writes on the bank's one stream.)
    kh_bank_execute is asynchronous and the two halves of consecutive launches overlap on two streams (double-buffered
    spectrum, IF-power and n0 partials; FM state by launch parity).  Six launches queued without a host sync in between
    must leave bit for bit the PCM, status and input status of six launches each followed by kh_bank_sync -- there are
    no atomics anywhere, so any difference is a hazard between launches."""
    fs, L, M, D, C, T = 8_000_000, 131072, 131073, 256, 8, 4
    N = L + M - 1
    calls = 6
    sigs = [(k - 4, 0.02 + 0.01 * k, 300.0 + 50 * k, 2500.0) for k in range(C)]
    x = np.ascontiguousarray(synth.wideband_fm(3, T * calls * L, fs, N, sigs), np.complex64)
    monkeypatch.setenv("KA9Q_HIP_PIPELINE", pipeline_env)  # read by kh_bank_create
    banks = []
    for _ in range(2):
        b = hip.Bank(L, M, D, 1, C, T, fs, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
        for c, (k, *_r) in enumerate(sigs):
            b.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0,
                          kaiser_beta=3.0)
        banks.append(b)
    a, q = banks
    # the queued bank takes its input from device memory (a device-to-device copy ordered on the bank's front stream)
    src = hip.Bank(L, M, D, 1, 1, T * calls, fs)  # only its raw ring is used: device-resident copy of the whole input
    src.write_input(0, x)
    xd, xcap = src.raw_device_ptr(0)
    assert xcap >= x.size
    for call in range(calls):
        seg = x[call * T * L:(call + 1) * T * L]
        a.write_input(0, seg)
        a.execute(T)
        a.sync()
    ptr, cap = q.raw_device_ptr(0)
    assert cap >= T * L
    for call in range(calls):
        # the raw ring is refilled by a device copy on the bank's front stream order: write_input with a device pointer
        q.write_input_device(0, xd + 8 * call * T * L, T * L)
        q.execute(T, sync=False)
    q.sync()
    for c in range(C):
        assert np.array_equal(a.pcm(c, T), q.pcm(c, T)), c
        assert a.status(c, T).tobytes() == q.status(c, T).tobytes(), c
    assert a.input_status(0, T).tobytes() == q.input_status(0, T).tobytes()
    a.close()
    q.close()
    src.close()


def test_full_ring_consumed_one_block_at_a_time(hip):
    """A latency-mode caller: T*L samples written at once, then consumed with execute(1) T times (every launch moves the
    ring's read offset; nothing is copied).  Every block must match the same
    samples run as one execute(T): PCM +-1 LSB, status 1e-4 -- and nothing may be processed twice."""
    fs, L, M, D, C, T = 1_000_000, 8192, 8193, 32, 3, 6
    N = L + M - 1
    sigs = [(-2, 0.05, 400.0, 2500.0), (0, 0.02, 700.0, 2000.0), (3, 0.1, 1000.0, 3000.0)]
    x = np.ascontiguousarray(synth.wideband_fm(5, T * L, fs, N, sigs), np.complex64)
    banks = []
    for _ in range(2):
        b = hip.Bank(L, M, D, 1, C, T, fs, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
        for c, (k, *_r) in enumerate(sigs):
            b.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0,
                          kaiser_beta=3.0)
        banks.append(b)
    whole, single = banks
    olen = L // D
    whole.write_input(0, x)
    whole.execute(T)
    single.write_input(0, x)
    for blk in range(T):
        single.execute(1)
        for c in range(C):
            sl = slice(blk * olen, (blk + 1) * olen)
            d = np.abs(single.pcm(c, 1).astype(np.int32) - whole.pcm(c, T)[sl].astype(np.int32))
            assert d.max() <= 1, (blk, c, int(d.max()))
            a, w = single.status(c, 1)[0], whole.status(c, T)[blk]
            for key in ("bb_power", "snr"):
                assert abs(a[key] - w[key]) <= 1e-4 * abs(w[key]) + 1e-4, (key, blk, c, a[key], w[key])
            assert a["squelch_open"] == w["squelch_open"]
    # the ring is empty now: one more launch must be refused, not run on stale samples
    with pytest.raises(Exception):
        single.execute(1)
    whole.close()
    single.close()


@pytest.mark.parametrize("payload_name", ["KH_PT_CF32", "KH_PT_IQ16LE"])
def test_ring_is_compacted_when_a_write_needs_the_room(hip, payload_name):
    """The raw ring keeps a read offset: a launch moves it, and the unread samples are moved to the start of the ring only
    when a write would run past the end -- by fewer samples than are left (several pieces, in ascending order) and by more
    (one piece).  A caller that tops the ring up between launches of different lengths must get what a caller gets that
    writes and consumes the same samples launch by launch: PCM +-1 LSB (launches of other lengths take other kernels),
    status 1e-4, over the whole sequence and with no sample processed twice or skipped."""
    fs, L, M, D, C, T = 1_000_000, 8192, 8193, 32, 3, 6
    N = L + M - 1
    payload = getattr(hip, payload_name)
    sigs = [(-2, 0.05, 400.0, 2500.0), (0, 0.02, 700.0, 2000.0), (3, 0.1, 1000.0, 3000.0)]
    total = 16
    x = np.ascontiguousarray(synth.wideband_fm(6, total * L, fs, N, sigs), np.complex64)
    if payload == hip.KH_PT_IQ16LE:
        raw = np.empty(2 * x.size, np.int16)
        raw[0::2] = np.clip(np.rint(x.real * 32767 * 4), -32768, 32767)
        raw[1::2] = np.clip(np.rint(x.imag * 32767 * 4), -32768, 32767)
        per = 2
    else:
        raw, per = x, 1
    banks = []
    for _ in range(2):
        b = hip.Bank(L, M, D, 1, C, T, fs, payload=payload, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
        for c, (k, *_r) in enumerate(sigs):
            b.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0,
                          kaiser_beta=3.0)
        banks.append(b)
    plain, ring = banks
    olen = L // D
    # (blocks written, then the launches that follow): the ring holds 6 blocks
    script = [(6, [1]),        # offset 1, 5 left
              (1, [1, 1]),     # 1 + 5 + 1 > 6: moved by 1 block in 5 pieces; then offset 2, 4 left
              (2, [1, 2, 1]),  # 2 + 4 + 2 > 6: moved by 2 blocks in 2 pieces; offset 4, 2 left
              (3, [2, 3]),     # 4 + 2 + 3 > 6: moved by 4 blocks in one piece; everything consumed: offset back to 0
              (4, [4])]        # a plain launch from the start of the ring again
    wpos = rpos = 0
    for nwrite, launches in script:
        ring.write_input(0, raw[wpos * L * per:(wpos + nwrite) * L * per], count=nwrite * L)
        wpos += nwrite
        for nb in launches:
            ring.execute(nb)
            plain.write_input(0, raw[rpos * L * per:(rpos + nb) * L * per], count=nb * L)
            plain.execute(nb)
            for c in range(C):
                d = np.abs(ring.pcm(c, nb).astype(np.int32) - plain.pcm(c, nb).astype(np.int32))
                assert d.max() <= 1, (rpos, nb, c, int(d.max()))
                a, w = ring.status(c, nb), plain.status(c, nb)
                for key in ("bb_power", "snr"):
                    assert np.all(np.abs(a[key] - w[key]) <= 1e-4 * np.abs(w[key]) + 1e-4), (key, rpos, c)
                assert np.array_equal(a["squelch_open"], w["squelch_open"])
            assert ring.input_status(0, nb).tobytes() == plain.input_status(0, nb).tobytes() or \
                np.allclose(ring.input_status(0, nb)["if_power"], plain.input_status(0, nb)["if_power"], rtol=1e-5)
            rpos += nb
    assert rpos == wpos == total
    assert np.abs(ring.pcm(0, 4)).max() > 100
    with pytest.raises(Exception):  # the ring is empty
        ring.execute(1)
    with pytest.raises(Exception):  # and it still refuses what does not fit
        ring.write_input(0, raw[:7 * L * per], count=7 * L)
    plain.close()
    ring.close()


@pytest.mark.parametrize("case", ["inband", "outband", "noise"])
@pytest.mark.parametrize("shape", [(1_000_000, 8192, 8193, 32), (8_000_000, 131072, 131073, 256)])
def test_compute_n0_modes_agree(hip, monkeypatch, shape, case):
    """compute_n0's second iteration (radio.c:451-462) has these routes for a four-step transform: the spectrum is read a
    second time (KA9Q_HIP_N0_DIRECT=0); the row pass takes it from its registers once the tiles of a block have exchanged
    their first-iteration sums (=1), or, when a tile's bounded wait for the others runs out, the block falls back to the
    re-read (forced here with KA9Q_HIP_N0_SPIN=0); and the bracket (=2, kh_n0.cuh): the row pass keeps the bins near an
    upper bound of the threshold and the exact threshold is applied to those afterwards, unless the bound is off by more
    than a quarter, which sends the block to the re-read.  The three inputs put the bracket through its paths: "inband"
    has its strongest carrier inside the excluded range (three quarters of the power: the bound is far off), "outband" has
    none there (the bound is tight, and the carriers lift the threshold far above the noise: no candidates), "noise" is
    an empty band (a tenth of the bins are candidates, more than the slots requested in advance).  Every route must
    give the float64 evaluation of the same definition to 1e-4 and the re-read's value to 1e-5, block after block (the
    smoothing of radio.c:199-202 carries across launches)."""
    fs, L, M, D = shape
    N = L + M - 1
    T, calls = 3, 2
    sigs = {"inband": [(-2, 0.05, 400.0, 2500.0), (0, 0.2, 700.0, 2000.0), (3, 0.1, 1000.0, 3000.0)],  # channel 0 sits in the n0 passband
            "outband": [(-2, 0.05, 400.0, 2500.0), (3, 0.1, 1000.0, 3000.0)], "noise": []}[case]
    x = np.ascontiguousarray(synth.wideband_fm(11, T * calls * L, fs, N, sigs), np.complex64)
    results = {}
    for name, env in (("reread", {"KA9Q_HIP_N0_DIRECT": "0"}), ("direct", {"KA9Q_HIP_N0_DIRECT": "1"}),
                      ("fallback", {"KA9Q_HIP_N0_DIRECT": "1", "KA9Q_HIP_N0_SPIN": "0"}), ("bracket", {"KA9Q_HIP_N0_DIRECT": "2"})):
        monkeypatch.delenv("KA9Q_HIP_N0_SPIN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)  # read by kh_bank_create
        bank = hip.Bank(L, M, D, 1, 1, T, fs, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
        bank.set_channel(0, demod=hip.KH_DEMOD_FM, rotate=0, low=-5500.0, high=5500.0, kaiser_beta=3.0)
        n0s, specs = [], []
        for call in range(calls):
            bank.write_input(0, x[call * T * L:(call + 1) * T * L])
            bank.execute(T)
            n0s += [float(v) for v in bank.input_status(0, T)["n0"]]
            specs += [bank.spectrum(0, b) for b in range(T)]
        bank.close()
        results[name] = n0s
        if name == "reread":
            ref, prev = [], None
            for sp in specs:
                v = _n0_float64(sp, N, -5500.0, 5500.0, fs)
                prev = v if prev is None else prev + .005 * (v - prev)
                ref.append(prev)
    for name, n0s in results.items():
        for b, (g, r) in enumerate(zip(n0s, ref)):
            assert abs(g - r) <= 1e-4 * r, (name, b, g, r)
    for name in ("direct", "fallback", "bracket"):
        for b, (g, r) in enumerate(zip(results[name], results["reread"])):
            assert abs(g - r) <= 1e-5 * abs(r), (name, b, g, r)


@pytest.mark.parametrize("shape", ["192k", "8M"])
def test_channel_state_moves_between_banks(hip, shape):
    """kh_bank_get_channel_state / kh_bank_set_channel_state: a channel's demodulator state (fm.c:26-33; linear.c:29-44 AGC, PLL,
    oscillators) carried to another bank that has been fed the same stream while tuned elsewhere.  After set_channel with the
    channel's parameters and set_channel_state, the next launch of the second bank equals the first bank's bit for bit --
    FM (de-emphasised and flat), USB with hang AGC, coherent AM with the PLL; the two banks have run different numbers of
    launches (different FM state parity)."""
    if shape == "192k":
        fs, L, M, D = 192_000, 3840, 4353, 4
        plan = [dict(demod=hip.KH_DEMOD_FM, k=-1), dict(demod=hip.KH_DEMOD_FM, k=2, flat=1),
                dict(demod=hip.KH_DEMOD_LINEAR, k=0, agc=1, hangtime=700.0, recovery_rate=10 ** ((20.0 / 48000) / 20), agc_gain=1e4, low=100.0, high=3000.0),
                dict(demod=hip.KH_DEMOD_LINEAR, k=3, pll=1, agc=1, hangtime=1.1 * 48000, recovery_rate=10 ** ((6.0 / 48000) / 20), agc_gain=1e4)]
    else:
        fs, L, M, D = 8_000_000, 131072, 131073, 256
        plan = [dict(demod=hip.KH_DEMOD_FM, k=-7), dict(demod=hip.KH_DEMOD_FM, k=4, flat=1), dict(demod=hip.KH_DEMOD_FM, k=9)]
    N = L + M - 1
    T = 3
    nblk = 3 * T
    t = np.arange(nblk * L) / fs
    x = synth.noise(nblk * L, 0.005, 9).astype(np.complex128)
    for j, ch in enumerate(plan):
        f = synth.rotate_for(ch["k"], 12500.0, N, fs) * fs / N
        if ch["demod"] == hip.KH_DEMOD_FM:
            x += 0.05 * np.exp(1j * (2 * np.pi * f * t + (2500.0 / (400 + 90 * j)) * np.sin(2 * np.pi * (400 + 90 * j) * t)))
        else:
            x += 0.05 * (1 + 0.5 * np.cos(2 * np.pi * 500 * t)) * np.exp(2j * np.pi * (f + 3.0) * t)
    x = x.astype(np.complex64)

    def tune(bank, c, ch, k):
        kw = {key: v for key, v in ch.items() if key not in ("demod", "k")}
        kw.setdefault("low", -5000.0)
        kw.setdefault("high", 5000.0)
        bank.set_channel(c, demod=ch["demod"], rotate=synth.rotate_for(k, 12500.0, N, fs), kaiser_beta=3.0, **kw)

    a = hip.Bank(L, M, D, 1, len(plan), T, fs)
    b = hip.Bank(L, M, D, 1, len(plan), T, fs)
    for c, ch in enumerate(plan):
        tune(a, c, ch, ch["k"])
        tune(b, c, ch, ch["k"] + 5)  # elsewhere on the raster
    for lo, nb in ((0, T), (T, T)):
        a.write_input(0, x[lo * L:(lo + nb) * L])
        a.execute(nb)
    for lo, nb in ((0, T), (T, 2), (T + 2, 1)):  # the same stream in three launches
        b.write_input(0, x[lo * L:(lo + nb) * L])
        b.execute(nb)
    assert hip.lib().kh_bank_channel_state_size() == len(a.get_channel_state(0))
    for c, ch in enumerate(plan):
        tune(b, c, ch, ch["k"])
        b.set_channel_state(c, a.get_channel_state(c))
    for bank in (a, b):
        bank.write_input(0, x[2 * T * L:3 * T * L])
        bank.execute(T)
    for c, ch in enumerate(plan):
        assert np.array_equal(a.pcm(c, T), b.pcm(c, T)), c
        assert a.status(c, T).tobytes() == b.status(c, T).tobytes(), c
        assert np.abs(a.pcm(c, T)).max() > 100, c
        assert a.get_channel_state(c) == b.get_channel_state(c), c  # (the record is normalised: slot 0 is what the next launch reads)
    a.close()
    b.close()
