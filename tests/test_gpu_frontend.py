"""GPU: front-end corrections (hackrf.c:822-919 rx_callback, funcube.c:364-428) and payload packing (hackrf.c:511-573)
against the oracle's restatement, through the C ABI.

The device takes each block's sums from exact integer moments; the reference (and the oracle) accumulate them in fp32
one sample at a time, which over 131072 samples carries ~1e-5 of rounding of its own.  Bars: estimator state (DC,
imbalance, sin phi, gains, in_power) 1e-4 relative (+1e-7 absolute for values near zero); corrected float rails 1e-5
relative RMS; funcube's int16 output within 1 LSB; rotation phase and clip count exact; packed payload bytes exact and
output energy 1e-4.  Parity unpinned (hackrf.c / funcube.c need libhackrf / portaudio headers)."""
import numpy as np
import pytest

import orc
from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _iq8(n, seed, dc=(3.0, -2.0), gain_q=0.9, phi=0.05, amp=40.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    ph = 2 * np.pi * 0.01234 * t + 0.3 * np.sin(2 * np.pi * 0.0007 * t)
    i = amp * np.cos(ph) + 6 * rng.standard_normal(n) + dc[0]
    q = gain_q * amp * np.sin(ph + phi) + 6 * rng.standard_normal(n) + dc[1]
    iq = np.empty(2 * n)
    iq[0::2], iq[1::2] = i, q
    iq = np.clip(np.round(iq), -128, 127).astype(np.int8)
    iq[10:2000:97] = -128  # clips on both rails (hackrf.c:841-848)
    return iq


def _check_state(got, ref):
    for k, v in ref.items():
        g = got[k][0]
        if k in ("rotate_phase", "clips"):
            assert int(g) == v, (k, g, v)
        else:
            assert abs(float(g) - v) <= 1e-4 * abs(v) + 1e-7, (k, g, v)


@pytest.mark.parametrize("samples,nblocks,offset,imb0", [(131072, 3, 1, 0.0), (4099, 7, 3, 1.0), (1000, 12, 0, 1.0),
                                                         (5, 2, -1, 1.0)])
def test_hackrf_correct_matches_oracle(hip, samples, nblocks, offset, imb0):
    iq = _iq8(samples * nblocks, seed=samples)
    rate = 1.0 / (10e6 * 1.0)  # hackrf.c:277 at 10 MS/s
    st_o = orc.fe_new(imb0)
    oi, oq = orc.hackrf_rx(st_o, iq, samples, nblocks, offset, rate)
    st_g = hip.fe_state(imb0)
    # two calls: the state carries over exactly as between reference callbacks
    k = nblocks // 2
    gi1, gq1 = hip.hackrf_correct(st_g, iq[:2 * samples * k], samples, k, offset, rate) if k else (np.empty(0, np.float32),) * 2
    gi2, gq2 = hip.hackrf_correct(st_g, iq[2 * samples * k:], samples, nblocks - k, offset, rate)
    gi, gq = np.concatenate([gi1, gi2]), np.concatenate([gq1, gq2])
    # 131072-sample transfers: the fp32 running sums of the reference carry ~2e-5 of their own rounding into
    # i_energy/q_energy, hence ~1e-5 into gain_i = sqrt((1 + 1/imbalance)/2) while imbalance is still small
    bar = 1e-5 if samples <= 16384 else 5e-5
    assert rel_rms(gi, oi) < bar and rel_rms(gq, oq) < bar
    _check_state(st_g, orc.fe_fields(st_o))
    assert st_g["clips"][0] > 0 or samples < 100


@pytest.mark.parametrize("blocksize,nblocks", [(240, 50), (1000, 4)])
def test_funcube_correct_matches_oracle(hip, blocksize, nblocks):
    rng = np.random.default_rng(blocksize)
    n = blocksize * nblocks
    t = np.arange(n)
    buf = np.empty(2 * n)
    buf[0::2] = 9000 * np.cos(2 * np.pi * 0.013 * t) + 300 * rng.standard_normal(n) + 120
    buf[1::2] = 8000 * np.sin(2 * np.pi * 0.013 * t + 0.04) + 300 * rng.standard_normal(n) - 75
    buf = np.round(buf).astype(np.int16)
    rate = blocksize / (192000.0 * 1.0)  # funcube.c:340
    st_o = orc.fe_new(1.0)
    ref = orc.funcube_blocks(st_o, buf, blocksize, nblocks, rate)
    st_g = hip.fe_state(1.0)
    got = hip.funcube_correct(st_g, buf, blocksize, nblocks, rate)
    d = np.abs(got.byteswap().astype(np.int32) - ref.byteswap().astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    _check_state(st_g, orc.fe_fields(st_o))


@pytest.mark.parametrize("pt", [97, 95, 98])
def test_hackrf_pack_matches_oracle(hip, pt):
    rng = np.random.default_rng(pt)
    n = 3001
    ip, qp = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32)
    atten = (127.0 if pt == 98 else 32767.0) * 0.5 ** 6 * 50  # Filter_atten (hackrf.c:286-289) times a signal level
    ip[:4] = [2.0, -2.0, 0.0, 1e-9]                           # the first two saturate
    ref, e_ref = orc.hackrf_pack(pt, ip, qp, atten)
    got, e_got = hip.hackrf_pack(pt, ip, qp, atten)
    assert np.array_equal(got, ref)
    assert abs(e_got - e_ref) <= 1e-4 * e_ref  # the reference's running fp32 sum over 6002 terms vs a tree sum


def test_frontend_error_behaviour(hip):
    st = hip.fe_state()
    assert hip.lib().kh_hackrf_correct(st.ctypes.data, None, 10, 1, 1, 1e-7, None, None) == -1
    x = np.zeros(4, np.float32)
    out = np.zeros(16, np.uint8)
    assert hip.lib().kh_hackrf_pack(10, x.ctypes.data, x.ctypes.data, 4, 1.0, out.ctypes.data, None) == -1
    assert "payload" in hip.last_error()
