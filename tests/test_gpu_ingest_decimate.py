"""GPU parity: proc_samples' convert + NCO mix + IF power (radio.c:112-204, osc.c:22-58) inside the bank, and the
half-band decimators (decimate.c:113-163; call pattern hackrf.c:495-509), against the oracle.

Integer unpacking is bit-exact by construction (same integer expression, one float multiply); the NCO is fp64 closed
form on the GPU against the oracle's fp64 recurrence, so the mixed samples agree to float rounding (1e-6 relative)."""
import ctypes as C

import numpy as np
import pytest

import orc
import pipeline
import synth
from conftest import rel_rms

pytestmark = pytest.mark.gpu

FS, L, M, D = 192_000, 3840, 4353, 4
N = L + M - 1


def _raw(payload, x, rng):
    """wire bytes for `payload` carrying (a quantised version of) the cf32 samples x."""
    if payload == orc.IQ_PT:
        return synth.to_int16(x)
    if payload == orc.PCM_STEREO_PT:
        return synth.to_int16(x).byteswap()
    if payload == orc.IQ_PT8:
        v = np.empty(2 * x.size, np.float32)
        v[0::2], v[1::2] = x.real, x.imag
        return np.clip(np.rint(v * 127), -128, 127).astype(np.int8)
    if payload == orc.IQ_PT12:
        return rng.integers(0, 256, 3 * x.size, dtype=np.uint8)  # any bit pattern is a valid 12-bit pair
    raise ValueError(payload)


@pytest.mark.parametrize("payload", [orc.IQ_PT, orc.PCM_STEREO_PT, orc.IQ_PT8, orc.IQ_PT12])
def test_wire_formats_through_the_bank(hip, payload):
    """raw payload -> float * gain_factor -> spectrum: the spectrum of block 0 and 1 and the IF power must match."""
    rng = np.random.default_rng(payload)
    T = 2
    x = (synth.noise(T * L, 0.2, 7) + synth.fm_carrier(0, T * L, FS, 20e3, 0.3, 700.0, 2500.0)).astype(np.complex64)
    raw = _raw(payload, x, rng)
    bank = hip.Bank(L, M, D, 1, 1, T, FS, payload=payload)
    pipe = pipeline.Pipeline(L, M, D, FS, payload=payload)
    bank.set_gain(0, 0.75)
    pipe.fins[0].gain = 0.75
    bank.set_channel(0, low=-5000.0, high=5000.0)
    pipe.add_channel(low=-5000.0, high=5000.0)
    per = raw.size // T
    bank.write_input(0, raw, count=T * L)
    bank.execute(T)
    ist = bank.input_status(0, T)
    for b in range(T):
        outs, istat = pipe.run_block([raw[b * per:(b + 1) * per]])
        assert rel_rms(bank.spectrum(0, b), pipe.spectrum(0)) < 1e-5, b
        assert abs(ist[b]["if_power"] - istat[0][0]) <= 1e-5 * istat[0][0]
        assert rel_rms(bank.baseband(0, T)[b * (L // D):(b + 1) * (L // D)], outs[0][0]) < 1e-5
    bank.close()


def test_exact_integer_unpack(hip):
    """With unit gain and no NCO the time samples are exactly int * (1/32767): check through an all-pass view --
    the forward transform of a single nonzero sample has constant magnitude |x|."""
    T = 1
    for payload, mk in [(orc.IQ_PT, lambda: np.array([12345, -32768], np.int16)),
                        (orc.PCM_STEREO_PT, lambda: np.array([12345, -32768], np.int16).byteswap()),
                        (orc.IQ_PT8, lambda: np.array([100, -128], np.int8))]:
        one = mk()
        raw = np.zeros(2 * L, one.dtype)
        raw[0:2] = one
        bank = hip.Bank(L, M, D, 1, 1, T, FS, payload=payload)
        bank.set_channel(0)
        bank.write_input(0, raw, count=L)
        bank.execute(1)
        spec = bank.spectrum(0, 0)
        scale = np.float32(1 / 127) if payload == orc.IQ_PT8 else np.float32(1 / 32767)
        vals = (np.array([12345, -32768]) if payload != orc.IQ_PT8 else np.array([100, -128])).astype(np.float32) * scale
        want = abs(complex(vals[0], vals[1]))
        assert np.allclose(np.abs(spec), want, rtol=2e-6), payload
        bank.close()


def test_lo2_and_doppler_sweep_phase_continuity_across_launches(hip):
    """radio.c:180-191 with set_doppler's sweep (osc.c:43-44): 6 launches of 3 blocks (crossing Renorm_rate = 16384
    samples several times), a retune of LO2 in mid-run (phase kept, osc.c:24-27) and a zero-filled gap
    (radio.c:112-131: the oscillators keep stepping)."""
    T, calls = 3, 6
    lo2 = -48_000.0 / FS
    dop_f, dop_r = 5_000.0 / FS, (-180.0 / FS) / FS
    x = (synth.noise(T * calls * L, 0.05, 9)
         + synth.swept_carrier(0, T * calls * L, FS, 48_000.0 - 5_000.0, 180.0, 0.2)).astype(np.complex64)
    bank = hip.Bank(L, M, D, 1, 1, T, FS)
    pipe = pipeline.Pipeline(L, M, D, FS)
    bank.set_channel(0, low=-5000.0, high=5000.0)
    pipe.add_channel(low=-5000.0, high=5000.0)
    bank.set_lo(0, lo2, dop_f, dop_r)
    pipe.set_lo(0, lo2, dop_f, dop_r)
    olen = L // D
    zero_blocks = {7}  # one lost block in the third launch
    for call in range(calls):
        if call == 4:  # retune LO2 by +1 kHz, keep Doppler: phase continuous.  pipe.set_lo would re-arm the sweep
            # from its start; the reference's set_second_LO (radio.c:350) touches only the LO2 oscillator
            new_lo = lo2 + 1000.0 / FS
            bank.set_second_lo(0, new_lo)
            orc.lib().orc_set_osc(C.addressof(pipe.fins[0].lo2), new_lo, 0.0)
        seg = x[call * T * L:(call + 1) * T * L].copy()
        for b in range(T):
            if call * T + b in zero_blocks:
                seg[b * L:(b + 1) * L] = 0
        for b in range(T):
            if call * T + b in zero_blocks:
                bank.write_zeros(0, L)
            else:
                bank.write_input(0, seg[b * L:(b + 1) * L])
        bank.execute(T)
        bb = bank.baseband(0, T)
        ist = bank.input_status(0, T)
        for b in range(T):
            outs, istat = pipe.run_block([seg[b * L:(b + 1) * L]])
            y = outs[0][0]
            assert rel_rms(bb[b * olen:(b + 1) * olen], y) < 1e-5, (call, b)
            assert abs(ist[b]["if_power"] - istat[0][0]) <= 1e-5 * istat[0][0] + 1e-12
    # the swept carrier was tracked open-loop: after the filter has filled, the baseband carrier is within a few Hz of DC
    ph = np.unwrap(np.angle(bb[olen:]))
    assert abs((ph[-1] - ph[0]) / (2 * np.pi) / (ph.size / 48000.0) + 0.0) < 1100.0  # (retuned by 1 kHz at call 4)
    bank.close()



# ------------------------------------------------------------------ large transforms: ingest fused into the FFT
@pytest.mark.parametrize("payload", [orc.IQ_PT, orc.PCM_STEREO_PT, orc.IQ_PT8, orc.IQ_PT12, 0])
@pytest.mark.parametrize("shape", [(8192, 8193, 16, 2), (4096, 12289, 16, 1)])
def test_fused_ingest_large_transform(hip, payload, shape):
    """N = 16384 goes through the four-step transform whose first pass reads the raw ring itself (convert, gain, IF
    power, overlap-save history): three launches so that the history crosses launches -- with L < M-1 and one block
    per launch the history even feeds the next history -- then a launch with LO2 set, which takes the unfused
    kh_ingest path and must continue from the same history."""
    Lb, Mb, Db, T = shape
    rng = np.random.default_rng(payload + Lb)
    calls = 4
    # (the carrier sits inside the passband: against an out-of-band one the output would be float round-off)
    x = (synth.noise(calls * T * Lb, 0.05, 11) + synth.fm_carrier(0, calls * T * Lb, FS, 1.5e3, 0.3, 700.0, 1000.0)).astype(np.complex64)
    raw = x if payload == 0 else _raw(payload, x, rng)
    per = raw.size // (calls * T)
    bank = hip.Bank(Lb, Mb, Db, 1, 1, T, FS, payload=payload)
    pipe = pipeline.Pipeline(Lb, Mb, Db, FS, payload=payload)
    bank.set_gain(0, 0.75)
    pipe.fins[0].gain = 0.75
    bank.set_channel(0, low=-4000.0, high=4000.0)
    pipe.add_channel(low=-4000.0, high=4000.0)
    olen = Lb // Db
    for call in range(calls):
        if call == 3:
            bank.set_second_lo(0, -1000.0 / FS)
            orc.lib().orc_set_osc(C.addressof(pipe.fins[0].lo2), -1000.0 / FS, 0.0)
        seg = raw[call * T * per:(call + 1) * T * per]
        bank.write_input(0, seg, count=T * Lb)
        bank.execute(T)
        ist = bank.input_status(0, T)
        bb = bank.baseband(0, T)
        for b in range(T):
            outs, istat = pipe.run_block([seg[b * per:(b + 1) * per]])
            assert rel_rms(bank.spectrum(0, b), pipe.spectrum(0)) < 1e-5, (call, b)
            # status-scalar bar (1e-4): the reference adds L floats one by one, the GPU in a tree
            assert abs(ist[b]["if_power"] - istat[0][0]) <= 1e-4 * istat[0][0], (call, b)
            y, want = bb[b * olen:(b + 1) * olen], outs[0][0]
            if (call * T + b) * Lb >= Mb - 1:
                assert rel_rms(y, want) < 1e-5, (call, b)
            else:  # filter still filling: the output is pre-ringing far below the carrier (0.3 * 0.75), bar relative to that
                assert np.linalg.norm(y - want) / np.sqrt(olen) < 1e-5 * 0.225, (call, b)
    bank.close()

# ------------------------------------------------------------------ decimate.c
def _f8_states(n):
    arr = (hip_Hb15 * n)()
    for s in arr:
        for j, c in enumerate((-6.0, 33.0, -116.0, 490.0)):
            s.coeffs[j] = c / 802.0
    return arr


hip_Hb15 = None


def test_hb15_and_hb3_block(hip):
    global hip_Hb15
    hip_Hb15 = hip.Hb15State
    rng = np.random.default_rng(12)
    Lh, Lo = hip.lib(), orc.lib()
    sg = _f8_states(1)
    so = orc.Hb15()
    Lo.orc_hb15_init_f8(C.addressof(so))
    h3g, h3o = C.c_float(0.0), C.c_float(0.0)
    for cnt in [1000, 7, 4096, 1, 300]:
        x = rng.uniform(-1, 1, 2 * cnt).astype(np.float32)
        yg, yo = np.empty(cnt, np.float32), np.empty(cnt, np.float32)
        xg = x.copy()  # (a bare x.copy().ctypes.data would free the copy before the call)
        Lh.hb15_block(sg, yg.ctypes.data, xg.ctypes.data, cnt)
        Lo.orc_hb15_block(C.addressof(so), yo.ctypes.data_as(C.c_void_p), x.copy().ctypes.data_as(C.c_void_p), cnt)
        assert np.abs(yg - yo).max() <= 2e-6, cnt
        # the carried state is copied samples: exact (slot 0 of even/odd is scratch, see test_oracle_golden)
        assert list(sg[0].even_samples)[1:] == list(so.even_samples)[1:]
        assert list(sg[0].odd_samples)[1:] == list(so.odd_samples)[1:]
        assert list(sg[0].old_odd_samples) == list(so.old_odd_samples)
        Lh.hb3_block(C.addressof(h3g), yg.ctypes.data, xg.ctypes.data, cnt)
        Lo.orc_hb3_block(C.addressof(h3o), yo.ctypes.data_as(C.c_void_p), x.copy().ctypes.data_as(C.c_void_p), cnt)
        assert np.abs(yg - yo).max() <= 1e-6
        assert h3g.value == h3o.value
    # in place, as hackrf.c:497-509 calls it
    x = rng.uniform(-1, 1, 2048).astype(np.float32)
    a, b = x.copy(), x.copy()
    Lh.hb15_block(sg, a.ctypes.data, a.ctypes.data, 1024)
    Lo.orc_hb15_block(C.addressof(so), b.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 1024)
    assert np.abs(a[:1024] - b[:1024]).max() <= 2e-6


@pytest.mark.parametrize("logd,thresh,chunk", [(6, 8, 480), (3, 8, 1000), (10, 8, 60), (1, 8, 4096), (6, 8, 5000), (6, 3, 517),
                                               (9, 8, 37), (8, 2, 100), (6, 8, 1), (4, 8, 5), (7, 0, 300)])
def test_decimation_cascade_rails(hip, logd, thresh, chunk):
    """hackrf.c:495-509: both rails through log2(D) in-place stages, hb3 for stages j >= stage_threshold (D >= 512),
    hb15 below; three consecutive chunks to exercise every stage's state carry.  64:1 at chunk 480 is the
    12.288 MS/s -> 192 kS/s case the DCC paper quotes."""
    global hip_Hb15
    hip_Hb15 = hip.Hb15State
    rng = np.random.default_rng(logd)
    Lh, Lo = hip.lib(), orc.lib()
    rails = 2
    n = chunk << logd
    sg = _f8_states(rails * logd)
    h3g = np.zeros(rails * logd, np.float32)
    so = [(orc.Hb15 * logd)() for _ in range(rails)]
    for r in range(rails):
        for j in range(logd):
            Lo.orc_hb15_init_f8(C.addressof(so[r][j]))
    h3o = [np.zeros(logd, np.float32) for _ in range(rails)]
    for _rep in range(3):
        x = rng.uniform(-1, 1, (rails, n)).astype(np.float32)
        g = x.copy()
        assert Lh.kh_decimate_rails(sg, h3g.ctypes.data, logd, thresh, g.ctypes.data, n, rails, chunk) == 0, hip.last_error()
        for r in range(rails):
            o = x[r].copy()
            Lo.orc_decimate_rail(so[r], h3o[r].ctypes.data_as(C.c_void_p), logd, thresh, o.ctypes.data_as(C.c_void_p), chunk)
            scale = np.abs(o[:chunk]).max()
            assert np.abs(g[r, :chunk] - o[:chunk]).max() <= 1e-5 * scale, (r, _rep)


@pytest.mark.parametrize("logd,thresh,chunks", [(6, 8, [480, 1, 3, 129, 1000]), (5, 2, [64, 7, 2000]), (9, 8, [16, 17, 5]),
                                                (2, 8, [3000, 2, 9000])])
def test_fused_cascade_equals_the_staged_one(hip, monkeypatch, logd, thresh, chunks):
    """The one-launch cascade (kh_decimate_fused: LDS tiles with halos, state from the last tile) against one launch pair
    per stage (KA9Q_HIP_DECIMATE_FUSED=0), both with the formulas of decimate.c:124-147,150-163: every output and every
    carried state bit for bit, over consecutive calls of different lengths (tiles that start in the carried history, a last
    tile shorter than the history, one-sample calls)."""
    global hip_Hb15
    hip_Hb15 = hip.Hb15State
    rng = np.random.default_rng(100 + logd)
    Lh = hip.lib()
    rails = 2
    sa, sb = _f8_states(rails * logd), _f8_states(rails * logd)
    ha, hb = np.zeros(rails * logd, np.float32), np.zeros(rails * logd, np.float32)
    for chunk in chunks:
        n = chunk << logd
        x = rng.uniform(-1, 1, (rails, n)).astype(np.float32)
        a, b = x.copy(), x.copy()
        monkeypatch.setenv("KA9Q_HIP_DECIMATE_FUSED", "1")
        assert Lh.kh_decimate_rails(sa, ha.ctypes.data, logd, thresh, a.ctypes.data, n, rails, chunk) == 0, hip.last_error()
        monkeypatch.setenv("KA9Q_HIP_DECIMATE_FUSED", "0")
        assert Lh.kh_decimate_rails(sb, hb.ctypes.data, logd, thresh, b.ctypes.data, n, rails, chunk) == 0, hip.last_error()
        assert np.array_equal(a[:, :chunk], b[:, :chunk]), chunk
        assert np.array_equal(a[:, chunk:], x[:, chunk:]), chunk  # the fused route leaves the rest of the caller's rails alone
        assert bytes(sa) == bytes(sb), chunk
        assert np.array_equal(ha, hb), chunk


def test_decimation_of_device_rails_reads_them_in_place(hip):
    """kh_decimate_rails with DEVICE pointers (hackrf.c:495-509 when kh_hackrf_correct has left the rails on the GPU): the fused
    cascade reads the caller's rails where they lie and writes the first `chunk` floats of each back; same floats as the
    host-pointer call, rail stride larger than the rail, the rest of the buffer untouched, states (host memory) carried."""
    from test_zz_no_alloc import Dev
    global hip_Hb15
    hip_Hb15 = hip.Hb15State
    rng = np.random.default_rng(77)
    Lh = hip.lib()
    rails, logd, thresh = 2, 6, 8
    stride = (3000 << logd) + 192  # floats between rails: more than any call's rail
    sa, sb = _f8_states(rails * logd), _f8_states(rails * logd)
    ha, hb = np.zeros(rails * logd, np.float32), np.zeros(rails * logd, np.float32)
    for chunk in (480, 3000, 129):
        n = chunk << logd
        x = np.zeros((rails, stride), np.float32)
        x[:, :n] = rng.uniform(-1, 1, (rails, n))
        x[:, n:] = 7.0  # must survive
        h = x.copy()
        assert Lh.kh_decimate_rails(sa, ha.ctypes.data, logd, thresh, h.ctypes.data, stride, rails, chunk) == 0, hip.last_error()
        d = Dev(x)
        assert Lh.kh_decimate_rails(sb, hb.ctypes.data, logd, thresh, d.ptr, stride, rails, chunk) == 0, hip.last_error()
        g = d.get()
        d.free()
        assert np.array_equal(g[:, :chunk], h[:, :chunk]), chunk
        assert np.array_equal(g[:, chunk:], x[:, chunk:]), chunk
        assert bytes(sa) == bytes(sb) and np.array_equal(ha, hb), chunk
