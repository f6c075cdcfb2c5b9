"""iqrecord/iqplay file format (ka9q-radio_amd/iqfile.py): CPU round trip of samples and attributes; on the GPU a
recorded file is replayed through the bank (payload IQ_PT samples, s16le) and the PCM must equal what the oracle's
chain makes of the same file: the file-based end-to-end test of SURVEY 8(f) rank 4."""
import os

import numpy as np
import pytest

import iqfile
import orc
import pipeline
import synth


def _record(tmp_path, n, fs=192_000):
    x = synth.noise(n, 0.01, 3) + synth.fm_carrier(0, n, fs, 0.0, 0.5, 1000.0, 3000.0)
    path = str(tmp_path / "iq-test.s16")
    attrs = iqfile.write(path, synth.to_int16(x.astype(np.complex64)), fs, 147_435_000.0, ssrc=0xBEEF01,
                         source_timestamp=1_234_567_890_123_456_789, source="192.0.2.1", multicast="iq.hf.mcast.local")
    return path, attrs


def test_round_trip_and_attributes(tmp_path):
    path, attrs = _record(tmp_path, 10_000)
    assert os.path.getsize(path) == 10_000 * 4                       # headerless, 2 x int16 per sample
    assert attrs["sampleformat"] == "s16le" and attrs["ssrc"] == "beef01" and attrs["frequency"] == "147435000.000"
    data, meta = iqfile.open_iq(path)
    assert data.shape == (10_000, 2) and data.dtype == np.dtype("<i2")
    assert meta == {"samprate": 192_000, "frequency": 147_435_000.0, "ssrc": 0xBEEF01, "sampleformat": "s16le",
                    "timestamp": 1_234_567_890_123_456_789}
    got = iqfile.read_attrs(path)
    assert got["source"] == "192.0.2.1" and got["multicast"] == "iq.hf.mcast.local" and got["channels"] == "2"
    # iqplay.c:196-201: without source_timestamp the start time comes from unixstarttime, moved to the GPS epoch in ns
    p2 = str(tmp_path / "b.s16")
    iqfile.write(p2, np.zeros(8, np.int16), 48000, 0.0, unixstarttime=1_546_300_800.25)
    _, m2 = iqfile.open_iq(p2)
    assert m2["timestamp"] == int((1_546_300_800.25 - 315964800 + 18) * 1e9) and m2["samprate"] == 48000
    # iqplay.c:254-260: power of a block; a short tail is not played
    blks = list(iqfile.blocks(data, 3840))
    assert len(blks) == 2
    v = np.asarray(data[:3840]).astype(np.float64)
    assert abs(blks[0][1] - (v * v).sum() / (32767.0 ** 2 * 3840)) < 1e-6 * blks[0][1]


@pytest.mark.gpu
def test_replay_through_the_bank_matches_oracle(hip, tmp_path):
    fs, L, M, D, T = 192_000, 3840, 4353, 4, 3
    nblocks = 7
    path, _ = _record(tmp_path, nblocks * L + 100)  # 100 samples of tail that iqplay would not send either
    bank = hip.Bank(L, M, D, 1, 1, T, fs, payload=hip.KH_PT_IQ16LE)
    kw = dict(rotate=0, low=-8000.0, high=8000.0, kaiser_beta=3.0, flat=0)
    bank.set_channel(0, demod=hip.KH_DEMOD_FM, **kw)
    pipe = pipeline.Pipeline(L, M, D, fs, payload=orc.IQ_PT)
    pipe.add_channel(demod="fm", **kw)
    data, meta = iqfile.open_iq(path)
    assert meta["samprate"] == fs
    olen = L // D
    launches = []

    def on_launch(first, nb):
        pcm = bank.pcm(0, nb)
        for b in range(nb):
            raw = np.ascontiguousarray(data[(first + b) * L:(first + b + 1) * L])
            outs, _ = pipe.run_block([raw])
            ref = orc.scaleclip(outs[0][1])
            d = np.abs(pcm[b * olen:(b + 1) * olen].astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 1, (first, b, d.max())
        launches.append(nb)

    _, done = iqfile.replay(path, bank, 0, T, on_launch)
    assert done == nblocks and launches == [3, 3, 1]
    assert np.abs(bank.pcm(0, 1)).max() > 1000  # there is audio
    bank.close()
