"""GPU: packet.c's receive side (kh_afsk_*: AFSK-1200 correlators, Gardner timing, NRZI/HDLC, crc_good) against the
oracle's restatement (oracle/ka9q_oracle.c: orc_afsk_block), through the C ABI.

Bars: decoded frames, their RTP sequence numbers/timestamps and the sample at which each frame closes are integer
work: bit-exact.  Bit times (sample index of every finished bit) bit-exact; the correlator output cur_val and the
analytic signal are fp32: 1e-5 relative RMS for the baseband, 1e-4 of the largest |cur_val| per bit.
Parity of the decoder itself is unpinned (packet.c does not build without FFTW, see tests/test_oracle_afsk.py);
crc_good is pinned to the reference."""
import numpy as np
import pytest

import orc
import synth
from conftest import rel_rms

pytestmark = pytest.mark.gpu
L = orc.AFSK_L


def _streams(nsess, nblocks, seed=0):
    """One PCM stream per session: different frames, amplitudes, noise levels and start offsets."""
    rng = np.random.default_rng(seed)
    xs, sent = [], []
    for s in range(nsess):
        frames = [synth.ax25_ui_frame("APRS", "KA9Q", b"s%d f%d " % (s, k) + bytes(rng.integers(0, 256, int(rng.integers(0, 11)), dtype=np.uint8)),
                                      src_ssid=s % 16) for k in range(int(rng.integers(1, 3)))]
        x = synth.afsk1200(synth.hdlc_bits(frames, preamble_flags=int(rng.integers(4, 10))), amp=float(rng.uniform(0.05, 0.8)),
                           lead=int(rng.integers(0, 3000)), tail=1000, phase0=float(rng.uniform(0, 6.28)))
        assert x.size <= nblocks * L, (x.size, nblocks * L)
        x = np.concatenate([x, np.zeros(nblocks * L - x.size)])
        # never digital silence: with an all-zero input the correlators hold FFT rounding residue of either sign, and
        # the bit clock (packet.c:467-469) would be steered by it -- differently on any two FFT implementations
        x = x + float(rng.choice([0.001, 0.003, 0.02])) * rng.standard_normal(x.size)
        if s % 5 == 4:
            x = -x
        xs.append(synth.pcm_be(x))
        sent.append(frames)
    return xs, sent


@pytest.mark.parametrize("split", [(40,), (1, 7, 32), (9, 9, 9, 9, 4)])
def test_afsk_sessions_match_oracle(hip, split):
    nsess, nblocks = 9, sum(split)
    xs, sent = _streams(nsess, nblocks, seed=len(split))
    dev = hip.Afsk(nsess, max(split), max_frames=8, max_bits=max(split) * 26)
    ors = [orc.AfskSession() for _ in range(nsess)]
    got_frames = [[] for _ in range(nsess)]
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
            assert rel_rms(dev.baseband(s, T), np.concatenate(yo)) < 1e-5, (s, pos)
            gb = dev.bits(s)
            assert [int(v) for v in gb["sample"]] == [b[0] for b in bo], (s, pos)
            ref = np.array([b[1] for b in bo], np.float32)
            assert np.abs(gb["cur_val"] - ref).max() <= 1e-4 * np.abs(ref).max(), (s, pos)
            gf = dev.frames(s)
            assert gf == fo, (s, pos)
            got_frames[s] += gf
            st = dev.status(s)
            assert (int(st["seq"]), int(st["timestamp"]), int(st["packets"]), int(st["bytes"])) == \
                (ors[s].st.out_seq, ors[s].st.out_timestamp, ors[s].st.out_packets, ors[s].st.out_bytes)
            assert int(st["flagsync"]) == ors[s].st.flagsync and int(st["frames_last_launch"]) == len(fo)
        pos += T
    for s in range(nsess):
        assert [f["data"] for f in got_frames[s]] == sent[s], s  # and the decoder really decodes
        ors[s].close()
    dev.close()


def test_afsk_error_behaviour(hip):
    assert not hip.lib().kh_afsk_create(0, 1, 1, 0)
    dev = hip.Afsk(2, 2, max_frames=1)
    dev.write_pcm(0, np.zeros(L, np.int16))
    with pytest.raises(RuntimeError, match="exactly"):
        dev.execute(1)                       # session 1 has nothing pending
    with pytest.raises(RuntimeError, match="exceed"):
        dev.write_pcm(0, np.zeros(2 * L, np.int16))
    with pytest.raises(RuntimeError, match="max_bits"):
        dev.bits(0)
    dev.close()
    assert hip.lib().kh_afsk_destroy(None) == 0
