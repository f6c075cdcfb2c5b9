"""The PCM packetiser restatements agree with each other: orc.RtpPacketiser (integer logic over PCM words, what the GPU
test compares the device plan with) against orc_packetise in oracle/ka9q_oracle.c (audio.c:32-135 over float audio)."""
import ctypes as C

import numpy as np
import pytest

import orc


@pytest.mark.parametrize("stereo", [0, 1])
def test_packetiser_restatements_agree(stereo):
    rng = np.random.default_rng(5 + stereo)
    chans = 2 if stereo else 1
    frames = 960
    st = orc.PcmState(0xFFFFFF00, 0xFFFD, 1, 0, 0, 0)
    py = orc.RtpPacketiser(chans, 0xFFFFFF00, 0xFFFD, 1)
    L = orc.lib()
    L.orc_packetise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for blk in range(12):
        audio = (rng.standard_normal(frames * chans) * 0.2).astype(np.float32)
        if blk % 3 == 1:
            audio[:] = 0  # a silent block
        if blk % 3 == 2:
            audio[:480] = 1e-6  # below one LSB: the first chunk is silent after scaleclip
        pcm_be = np.zeros(frames * chans, np.int16)
        sent, marker, ts = np.zeros(8, np.int32), np.zeros(8, np.int32), np.zeros(8, np.uint32)
        seq_before = st.seq
        n = L.orc_packetise(C.byref(st), audio.ctypes.data, frames, stereo, pcm_be.ctypes.data, sent.ctypes.data,
                            marker.ctypes.data, ts.ctypes.data, 8)
        words = orc.scaleclip(audio)
        assert np.array_equal(pcm_be.byteswap(), words)
        got = py.block(words)
        assert len(got) == n
        seq = seq_before
        for k, (t, s, m, snd, off, nw) in enumerate(got):
            assert (t, m if snd else 0, snd) == (int(ts[k]), int(marker[k]), int(sent[k]))
            if snd:
                assert s == seq
                seq = (seq + 1) & 0xFFFF
        assert (py.timestamp, py.seq, py.silent, py.packets, py.bytes, py.samples) == (
            st.timestamp, st.seq, st.silent, st.packets, st.bytes, st.samples)
