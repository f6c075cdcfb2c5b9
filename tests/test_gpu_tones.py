"""GPU: pl.c's CTCSS/DTMF correlator banks (kh_tones_*) against the oracle's restatement (orc_pl_samples), through the
C ABI.  Decisions (tone index, tone, digit, group indices) are integer work: exact.  Correlator energies are sums of
12000 / 2400 products that the reference accumulates in complex float, one rounding per sample (pl.c:253,267-268); the
device adds them in fp64 in another order: 1e-4 relative on the energies, 1e-3 dB on the PL ratio, 1e-3 relative on the
DTMF ratios.  Parity unpinned (pl.c does not build without the reference's socket stack)."""
import numpy as np
import pytest

import orc
import synth
from test_oracle_afsk import tone_audio

pytestmark = pytest.mark.gpu
W = 2400


@pytest.mark.parametrize("split", [(20,), (3, 5, 12), (1, 1, 1, 1, 1, 7, 8)])
def test_tones_match_oracle(hip, split):
    nwin = sum(split)
    cases = [dict(pl_hz=100.0, digits=[("5", 0.2, 0.4)]), dict(pl_hz=67.0, digits=[("1", 0.05, 0.3), ("D", 0.55, 0.8)]),
             dict(pl_hz=254.1, voice=0.4), dict(voice=0.3), dict(pl_hz=151.4, noise=0.05, digits=[("*", 0.0, 1.0)]),
             dict(pl_hz=150.0, digits=[("0", 0.31, 0.52), ("9", 0.6, 0.9)], noise=0.002)]
    xs = [synth.pcm_be(tone_audio(nwin * W, seed=k, **c)) for k, c in enumerate(cases)]
    xs.append(np.zeros(nwin * W, np.int16))  # digital silence: the NaN quirk of pl.c:402-409
    S = len(xs)
    dev = hip.Tones(S, max(split))
    ors = [orc.PlSessionRunner() for _ in range(S)]
    pos = 0
    seen_pl = 0
    for nw in split:
        for s in range(S):
            dev.write_pcm(s, xs[s][pos * W:(pos + nw) * W])
        dev.execute(nw)
        for s in range(S):
            pl, dt = ors[s].samples(xs[s][pos * W:(pos + nw) * W])
            gp, gd = dev.pl(s), dev.dtmf(s)
            assert len(gp) == len(pl) and len(gd) == len(dt) == nw
            for a, b in zip(gp, pl):
                assert a["tone"] == np.float32(b["tone"]), (s, pos, a, b)
                if np.isnan(b["snr"]):
                    assert np.isnan(a["snr"])
                    continue
                assert a["index"] == b["index"]
                assert abs(a["e0"] - b["e0"]) <= 1e-4 * b["e0"] and abs(a["e1"] - b["e1"]) <= 1e-4 * b["e1"]
                assert abs(a["snr"] - b["snr"]) <= 1e-3
            for a, b in zip(gd, dt):
                assert (chr(a["digit"]) if a["digit"] else "") == b["digit"], (s, pos, a, b)
                assert (a["low_index"], a["high_index"]) == (b["low_index"], b["high_index"])
                for k in ("low_snr", "high_snr"):
                    assert (np.isnan(a[k]) and np.isnan(b[k])) or abs(a[k] - b[k]) <= 1e-3 * abs(b[k]), (s, pos, k, a, b)
            seen_pl += len(pl)
        pos += nw
    assert seen_pl == S * (nwin // 5)
    dev.close()


def test_tones_error_behaviour(hip):
    assert not hip.lib().kh_tones_create(1, 0)
    dev = hip.Tones(2, 2)
    dev.write_pcm(0, np.zeros(W, np.int16))
    with pytest.raises(RuntimeError, match="exactly"):
        dev.execute(1)
    with pytest.raises(RuntimeError, match="exceed"):
        dev.write_pcm(0, np.zeros(2 * W, np.int16))
    dev.close()
    assert hip.lib().kh_tones_destroy(None) == 0
