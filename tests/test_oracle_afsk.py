"""CPU: the oracle's restatement of packet.c's AFSK-1200/HDLC decoder and ax25.c's crc_good.

crc_good is pinned to the REAL reference (ax25.c compiled unmodified into oracle/_ref): committed vectors in
tests/golden/ref_ax25.npz and, where oracle/_ref exists, live on random frames.  The decoder itself (packet.c:402-545)
includes filter.h -> <fftw3.h> and cannot be built here: its parity is unpinned; what is checked is that the
restatement decodes frames built independently (synth.ax25_ui_frame / hdlc_bits / afsk1200)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_crc_good_matches_reference_vectors():
    g = np.load(os.path.join(GOLD, "ref_ax25.npz"), allow_pickle=False)
    data, lens, good = g["data"], g["len"], g["good"]
    assert good.sum() >= 10 and (good == 0).sum() >= 10
    for row, n, want in zip(data, lens, good):
        assert orc.crc_good(row[:n].tobytes()) == bool(want)


@pytest.mark.skipif(orc.ref() is None or not hasattr(orc.ref(), "crc_good"), reason="oracle/_ref not built")
def test_crc_good_live_against_reference_random():
    R = orc.ref()
    rng = np.random.default_rng(11)
    for k in range(300):
        n = int(rng.integers(1, 400))
        body = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        f = body + synth.ax25_crc(body) if k % 2 else body + bytes(rng.integers(0, 256, 2, dtype=np.uint8))
        b = np.frombuffer(f, np.uint8).copy()
        assert bool(R.crc_good(b.ctypes.data_as(C.c_void_p), b.size)) == orc.crc_good(f)
        if k % 2:
            assert orc.crc_good(f)


def _decode(x, fast=False):
    s = orc.AfskSession(fast=fast)
    pcm = synth.pcm_be(x)
    frames, bits = [], []
    for b in range(pcm.size // orc.AFSK_L):
        _, fr, bt = s.block(pcm[b * orc.AFSK_L:(b + 1) * orc.AFSK_L])
        frames += fr
        bits += bt
    st = s.st
    out = dict(frames=frames, bits=bits, seq=st.out_seq, timestamp=st.out_timestamp, packets=st.out_packets,
               bytes=st.out_bytes)
    s.close()
    return out


def _signal(frames, amp=0.5, noise=0.0, seed=0, **kw):
    x = synth.afsk1200(synth.hdlc_bits(frames, **kw), amp=amp, lead=700, tail=3000)
    n = -(-x.size // orc.AFSK_L) * orc.AFSK_L
    x = np.concatenate([x, np.zeros(n - x.size)])
    if noise:
        x = x + noise * np.random.default_rng(seed).standard_normal(n)
    return x


def test_decodes_independently_built_frames():
    frames = [synth.ax25_ui_frame("APRS", "KA9Q", b"!4903.50N/07201.75W-Test 001234", src_ssid=7),
              synth.ax25_ui_frame("CQ", "N0CALL", bytes(range(256)), dest_ssid=1),   # every byte value: bit stuffing
              synth.ax25_ui_frame("BEACON", "W1AW", b"\xff" * 40 + b"\x7e\x7e\x00")]
    r = _decode(_signal(frames))
    assert [f["data"] for f in r["frames"]] == frames
    assert [f["seq"] for f in r["frames"]] == [0, 1, 2]          # packet.c:498
    ts = np.cumsum([0] + [len(f) for f in frames])
    assert [f["timestamp"] for f in r["frames"]] == list(ts[:-1])  # packet.c:500-501
    assert r["packets"] == 3 and r["bytes"] == sum(len(f) for f in frames) and r["timestamp"] == ts[-1]
    # one bit per 40 samples, give or take the Gardner steps (packet.c:453-469)
    d = np.diff([b[0] for b in r["bits"]])
    assert set(d.tolist()) <= {39, 40, 41}


def test_noise_polarity_and_damage():
    frames = [synth.ax25_ui_frame("APRS", "KA9Q", b"noisy but fine %d" % k) for k in range(4)]
    r = _decode(_signal(frames, amp=0.3, noise=0.03, seed=3))
    assert [f["data"] for f in r["frames"]] == frames
    # inverted audio: tone energies do not care
    r = _decode(-_signal(frames, amp=0.3))
    assert [f["data"] for f in r["frames"]] == frames
    # a damaged frame fails crc_good and is dropped; its neighbours survive and the sequence numbers stay dense
    bad = bytearray(frames[1])
    bad[20] ^= 0x10
    r = _decode(_signal([frames[0], bytes(bad), frames[2]]))
    assert [f["data"] for f in r["frames"]] == [frames[0], frames[2]]
    assert [f["seq"] for f in r["frames"]] == [0, 1]
    # seven ones in a row: abort (packet.c:529-535), then resynchronise on the next flag
    bits = synth.hdlc_bits([frames[0]])
    bits = bits[:200] + [1] * 9 + bits[200:] + synth.hdlc_bits([frames[3]])
    x = synth.afsk1200(bits, lead=100, tail=2000)
    x = np.concatenate([x, np.zeros(-x.size % orc.AFSK_L)])
    r = _decode(x)
    assert [f["data"] for f in r["frames"]] == [frames[3]]


def test_unsigned_ntohs_quirk():
    """packet.c:343: ntohs() is unsigned, so negative samples arrive offset by +2."""
    pcm = synth.pcm_be(np.array([0.25, -0.25, 0.0, -1.0 / 32767]))
    out = np.empty(4, np.float32)
    orc.lib().orc_pcm_to_real(pcm.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 4)
    assert np.allclose(out, [8191 / 32768, (65536 - 8191) / 32768, 0.0, 65535 / 32768], rtol=1e-7)


def test_fast_build_decodes_the_same_frames():
    frames = [synth.ax25_ui_frame("APRS", "KA9Q", b"abc %d" % k) for k in range(3)]
    x = _signal(frames, noise=0.02, seed=8)
    a, b = _decode(x), _decode(x, fast=True)
    assert [f["data"] for f in a["frames"]] == [f["data"] for f in b["frames"]] == frames


# ---------------------------------------------------------------- pl.c (unpinned: pl.c needs multicast.h's socket stack)
def tone_audio(n, pl_hz=None, digits=(), noise=0.01, seed=0, voice=0.2):
    """48 kS/s test audio: an optional CTCSS tone under 1 kHz 'voice', DTMF digits as (char, start s, stop s)."""
    fs = 48000.0
    t = np.arange(n) / fs
    x = voice * np.sin(2 * np.pi * 1000.0 * t) + noise * np.random.default_rng(seed).standard_normal(n)
    if pl_hz:
        x += 0.1 * np.sin(2 * np.pi * pl_hz * t)
    low = {"1": 697, "2": 697, "3": 697, "A": 697, "4": 770, "5": 770, "6": 770, "B": 770, "7": 852, "8": 852, "9": 852,
           "C": 852, "*": 941, "0": 941, "#": 941, "D": 941}
    high = {"1": 1209, "4": 1209, "7": 1209, "*": 1209, "2": 1336, "5": 1336, "8": 1336, "0": 1336, "3": 1477, "6": 1477,
            "9": 1477, "#": 1477, "A": 1633, "B": 1633, "C": 1633, "D": 1633}
    for d, a, b in digits:
        m = (t >= a) & (t < b)
        x[m] += 0.2 * np.sin(2 * np.pi * low[d] * t[m]) + 0.2 * np.sin(2 * np.pi * high[d] * t[m])
    return x


def test_pl_and_dtmf_decode_synthetic_audio():
    x = tone_audio(48000, pl_hz=100.0, digits=[("5", 0.5, 0.7), ("#", 0.8, 0.95)], voice=0.0)
    r = orc.PlSessionRunner()
    pl, dt = r.samples(synth.pcm_be(x))
    assert len(pl) == 4 and len(dt) == 20
    assert all(p["tone"] == 100.0 and p["index"] == 12 and p["snr"] >= 6 for p in pl)
    got = "".join(d["digit"] or "." for d in dt)
    assert got == "..........5555..###.", got
    # 48 of the 55 tones are found in clean audio and none is ever mistaken for another; the 250 ms interval cannot
    # separate neighbours closer than ~2.5 Hz by 6 dB (67.0/69.3, 150.0/151.4 ...: "some spectral leakage", pl.c:379)
    for hz in (103.5, 123.0, 254.1):
        pl, _ = orc.PlSessionRunner().samples(synth.pcm_be(tone_audio(24000, pl_hz=hz, voice=0.0)))
        assert [p["tone"] for p in pl] == [np.float32(hz)] * 2
    for hz in (67.0, 151.4):
        pl, _ = orc.PlSessionRunner().samples(synth.pcm_be(tone_audio(24000, pl_hz=hz, voice=0.0)))
        assert [p["tone"] for p in pl] == [0.0, 0.0] and all(p["snr"] < 6 for p in pl)
    pl, _ = orc.PlSessionRunner().samples(synth.pcm_be(tone_audio(24000, voice=0.3)))
    assert [p["tone"] for p in pl] == [0.0, 0.0]


def test_pl_digital_silence_quirk():
    """all-zero audio: 0/0 is NaN, `NaN < 10` is false, so process_dtmf keeps the last index of each group -> 'D'
    (pl.c:402-409); process_pl sees -inf - -inf = NaN and reports no tone."""
    pl, dt = orc.PlSessionRunner().samples(np.zeros(12000, np.int16))
    assert pl[0]["tone"] == 0.0 and np.isnan(pl[0]["snr"])
    assert [d["digit"] for d in dt] == ["D"] * 5
