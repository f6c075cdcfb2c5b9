"""GPU parity: the filter.h drop-in ABI of libka9q_hip against the CPU oracle (filter.c:57-702).

Bars (BASELINE.json north_star): bin/decimation indices bit-exact, complex baseband <= 1e-5 relative RMS.
"""
import numpy as np
import pytest

import orc
from conftest import rel_rms

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _noise(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(0.1)


# (3840, 3841, 4) is what the reference derives by default today: N = nextfastfft(2L-1) = 7680 = 2^9*3*5 (main.c:326-330,
# misc.c:129-148); (2800, 2801, 7) has factors 5 and 7; (960, 961, 1) is the 48 kS/s case without decimation
@pytest.mark.parametrize("L,M,D", [(3840, 4353, 4), (512, 513, 2), (131072, 131073, 256), (524288, 524289, 1024),
                                   (3840, 3841, 4), (2800, 2801, 7), (960, 961, 1)])
def test_forward_spectrum_and_output(hip, L, M, D):
    g = hip.Filter(L, M, D)
    o = orc.Filter(L, M, D)
    fs_out = 1.0
    g.set_filter(-0.17, 0.17, 3.0)
    o.set_filter(-0.17, 0.17, 3.0)
    assert rel_rms(g.response(), o.response()) < TOL
    assert abs(g.fout.contents.noise_gain - o.noise_gain()) <= 1e-5 * abs(o.noise_gain())
    nblocks = 3
    x = _noise(nblocks * L, 7)
    for b in range(nblocks):
        blk = x[b * L:(b + 1) * L]
        yg = g.execute(blk)
        yo = o.execute(blk)
        assert rel_rms(g.fdomain(), o.fdomain()) < TOL, f"spectrum block {b}"
        assert rel_rms(yg, yo) < TOL, f"output block {b}"
    g.close()
    o.close()


@pytest.mark.parametrize("rotate", [0, 2, -2, 100, -101, 4095, -4096, 8191, -8191])
def test_rotate_indexing_bit_exact(hip, rotate):
    """filter.c:274-289: with an all-ones response the output bins ARE the selected input bins, so an
    impulse spectrum exposes any off-by-one in the modular index arithmetic."""
    L, M, D = 3840, 4353, 4
    N, ND = L + M - 1, (L + M - 1) // D
    g = hip.Filter(L, M, D)
    o = orc.Filter(L, M, D)
    g.set_filter(-0.17, 0.17, 3.0)
    o.set_filter(-0.17, 0.17, 3.0)
    x = _noise(L, 11)
    assert rel_rms(g.execute(x, rotate), o.execute(x, rotate)) < TOL
    g.close()
    o.close()


@pytest.mark.parametrize("out_type", [orc.COMPLEX, orc.CROSS_CONJ])
@pytest.mark.parametrize("rotate", [0, 3, -700, 16383, -16383, 16384 - 512, 511 - 16384])
@pytest.mark.parametrize("D", [16, 32])
def test_wave_resident_output_rotate_and_cross_conj(hip, out_type, rotate, D):
    """N_dec = 1024 / olen = 512 (D = 16: one wavefront per block) and N_dec = 512 / olen = 256 (D = 32: half a
    wavefront per block, here with the other half idle) take the wave-resident kernel (kh_chan_kernels.hip): the same
    index arithmetic (filter.c:274-289, including rotations that wrap around N in the middle of the channel) and the
    CROSS_CONJ fold (filter.c:371-381), which that kernel does while loading."""
    L, M = 8192, 8193
    g = hip.Filter(L, M, D, orc.COMPLEX, out_type)
    o = orc.Filter(L, M, D, orc.COMPLEX, out_type)
    g.set_filter(-0.2, 0.31, 3.0)
    o.set_filter(-0.2, 0.31, 3.0)
    x = _noise(2 * L, 13)
    for b in range(2):
        blk = x[b * L:(b + 1) * L]
        assert rel_rms(g.execute(blk, rotate), o.execute(blk, rotate)) < TOL, b
    g.close()
    o.close()


@pytest.mark.parametrize("in_type,out_type", [(orc.COMPLEX, orc.CROSS_CONJ), (orc.COMPLEX, orc.REAL),
                                              (orc.REAL, orc.COMPLEX), (orc.REAL, orc.REAL)])
@pytest.mark.parametrize("shape", [(3840, 4353, 4), (3840, 3841, 4)])
def test_type_combinations(hip, in_type, out_type, shape):
    L, M, D = shape
    g = hip.Filter(L, M, D, in_type, out_type)
    o = orc.Filter(L, M, D, in_type, out_type)
    g.set_filter(0.01, 0.15, 3.0)
    o.set_filter(0.01, 0.15, 3.0)
    assert rel_rms(g.response(), o.response()) < TOL
    x = _noise(2 * L, 5)
    if in_type == orc.REAL:
        x = x.real.copy()
    for b in range(2):
        blk = x[b * L:(b + 1) * L]
        assert rel_rms(g.execute(blk, 3), o.execute(blk, 3)) < TOL
    g.close()
    o.close()


def test_kaiser_and_window_filter(hip):
    import ctypes as C
    for M, beta in [(1089, 3.0), (64, 0.0), (513, 7.5), (2, 1.0)]:
        wg = np.empty(M, np.float32)
        wo = np.empty(M, np.float32)
        assert hip.lib().make_kaiser(wg.ctypes.data, M, beta) == 0
        orc.lib().orc_make_kaiser(wo.ctypes.data_as(C.c_void_p), M, beta)
        assert np.allclose(wg, wo, rtol=2e-6, atol=1e-7), (M, beta)
    rng = np.random.default_rng(3)
    for Lw, Mw in [(960, 1089), (100, 157), (500, 3597)]:  # the last one exercises filter.c:543's in-place overlap
        N = Lw + Mw - 1
        r = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
        rg, ro = r.copy(), r.copy()
        assert hip.lib().window_filter(Lw, Mw, rg.ctypes.data, 3.0) == 0, hip.last_error()
        orc.lib().orc_window_filter(Lw, Mw, ro.ctypes.data_as(C.c_void_p), 3.0)
        assert rel_rms(rg, ro) < TOL, (Lw, Mw)
    # real filter
    N = 2048
    r = (rng.standard_normal(N // 2 + 1) + 1j * rng.standard_normal(N // 2 + 1)).astype(np.complex64)
    rg, ro = r.copy(), r.copy()
    assert hip.lib().window_rfilter(960, 1089, rg.ctypes.data, 3.0) == 0
    orc.lib().orc_window_rfilter(960, 1089, ro.ctypes.data_as(C.c_void_p), 3.0)
    assert rel_rms(rg, ro) < TOL


def test_error_behaviour(hip):
    L = hip.lib()
    assert L.execute_filter_input(None) == -1      # filter.c:214-215
    assert L.execute_filter_output(None, 0) == -1  # filter.c:242-243
    assert L.delete_filter_input(None) == 0        # filter.c:387-388
    assert L.delete_filter_output(None) == 0
    assert L.make_kaiser(None, 5, 1.0) == -1       # filter.c:483-484
    assert L.window_filter(10, 10, None, 1.0) == -1
    assert L.set_filter(None, 0.0, 0.1, 3.0) == -1  # filter.c:661-662
    assert not L.create_filter_output(None, None, 4, 1)  # filter.c:165-166
    assert np.isnan(L.noise_gain(None))            # filter.c:632-633
    assert not L.create_filter_input(1000, 1003, 1)  # N = 2002 = 2*7*11*13: a factor beyond 7 -> refused, with a message
    assert "power of two" in hip.last_error()
    g = hip.Filter(3840, 4353, 4)
    assert L.set_filter(g.fout, float("nan"), 0.1, 3.0) == -1
    assert np.isnan(g.fout.contents.noise_gain)    # filter.c:186-187: no response yet
    g.close()
