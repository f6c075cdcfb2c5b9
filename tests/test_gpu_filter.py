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
def test_rotated_output_matches_oracle(hip, rotate):
    """filter.c:274-289 with rotate != 0 (including the largest ones, which wrap around N) on noise through a Kaiser
    response: the rotated output against the oracle's, 1e-5 relative RMS.  (The exact statement about which input
    bin reaches which output bin is test_bin_map_is_exact below.)"""
    L, M, D = 3840, 4353, 4
    g = hip.Filter(L, M, D)
    o = orc.Filter(L, M, D)
    g.set_filter(-0.17, 0.17, 3.0)
    o.set_filter(-0.17, 0.17, 3.0)
    x = _noise(L, 11)
    assert rel_rms(g.execute(x, rotate), o.execute(x, rotate)) < TOL
    g.close()
    o.close()


def _bin_map(N, ND, rotate, p):
    """filter.c:276-289 as integers: the input bin m that output bin p is taken from.  p starts at N_dec/2+1 and m at
    N - N_dec/2 + 1 + rotate (ONE wrap into [0, N)), both step by one and wrap; so the Nyquist output bin N_dec/2 is
    reached last, i.e. from the positive side."""
    m = N - ND // 2 + 1 + rotate
    if m >= N:
        m -= N
    if m < 0:
        m += N
    i = (p - (ND // 2 + 1)) % ND
    return (m + i) % N


@pytest.mark.parametrize("out_type", [orc.COMPLEX, orc.CROSS_CONJ])
@pytest.mark.parametrize("shape", [(3840, 4353, 4), (8192, 8193, 16), (8192, 8193, 32)],
                         ids=["kh_filter_out-ND2048", "wave-ND1024", "halfwave-ND512"])
def test_bin_map_is_exact(hip, shape, out_type):
    """Which input bin reaches which output bin, as an exact statement (north_star: bit-exact bin/decimation indexing).
    The response is a delta at output bin p (caller-supplied through ka9q_hip_alloc / create_filter_output, as
    modulate.c:121-138 does); the input is a sum of exact on-bin tones, every bin near the expected one carrying its own
    amplitude code.  The output block is then a pure tone at decimated bin p whose amplitude names the ONE input bin
    that was selected: it must be the bin of filter.c:276-289's integer map, for p at both ends and around the Nyquist
    bin, and for rotations of 0, +-1, the largest legal ones and N/2 -- in the LDS-tile kernel (N_dec 2048), the
    wave-resident kernel (N_dec 1024) and its half-wave form (N_dec 512).  CROSS_CONJ (filter.c:371-381) folds the
    lone bin onto Q (p < N_dec/2: output purely imaginary) or I (p > N_dec/2: purely real) at twice the amplitude."""
    import ctypes as C
    L, M, D = shape
    N = L + M - 1
    ND = N // D
    lib = hip.lib()
    ps = [0, 1, ND // 2 - 1, ND // 2, ND // 2 + 1, ND - 1]
    nb = -(-N // L)  # blocks until the window holds N samples of the tones: from then on every bin is exact
    n = np.arange(nb * L, dtype=np.float64)
    for rotate in (0, 1, -1, N - 1, -(N - 1), N // 2, -(N // 2), N - ND // 2, ND // 2 - N + 3):
        want = {p: _bin_map(N, ND, rotate, p) for p in ps}
        bins = sorted({(m + d) % N for m in want.values() for d in (-1, 0, 1)})
        code = {m: 1 + k for k, m in enumerate(bins)}  # amplitude code of every bin near an expected one
        x = np.zeros(nb * L, np.complex128)
        for m, a in code.items():
            x += a * np.exp(2j * np.pi * ((m * n) % N) / N)
        x = (1e-3 * x).astype(np.complex64)
        fin = lib.create_filter_input(L, M, orc.COMPLEX)
        assert fin, hip.last_error()
        outs = {}
        for p in ps:
            resp = lib.ka9q_hip_alloc(ND * 8)
            assert resp
            r = hip._as_np(resp, ND, np.complex64)
            r[:] = 0
            r[p] = 1.0
            fo = lib.create_filter_output(fin, resp, D, out_type)  # the response is the filter's from here on
            assert fo, hip.last_error()
            outs[p] = fo
        inp = hip._as_np(fin.contents.input.c, L, np.complex64)
        for b in range(nb):
            inp[:] = x[b * L:(b + 1) * L]
            assert lib.execute_filter_input(fin) == 0
        for p, fo in outs.items():
            assert lib.execute_filter_output(fo, rotate) == 0, hip.last_error()
            olen = fo.contents.olen
            y = hip._as_np(fo.contents.output.c, olen, np.complex64).astype(np.complex128)
            amp = 1e-3 * N * code[want[p]]  # |X[m]| of the expected bin; its neighbours differ by 1e-3 * N
            where = (shape, out_type, rotate, p, want[p])
            if out_type == orc.COMPLEX or p in (0, ND // 2):
                assert np.allclose(np.abs(y), amp, rtol=2e-4), (where, float(np.abs(y).mean()) / (1e-3 * N))
                step = np.angle(np.mean(y[1:] * np.conj(y[:-1])))  # the output tone sits at decimated bin p
                assert abs(np.angle(np.exp(1j * (step - 2 * np.pi * p / ND)))) < 1e-4, (where, step)
            else:
                live, dead = (y.imag, y.real) if p < ND // 2 else (y.real, y.imag)
                assert np.abs(dead).max() < 2e-4 * amp, (where, float(np.abs(dead).max()))
                assert abs(np.abs(live).max() - 2 * amp) < 1e-2 * amp, (where, float(np.abs(live).max()) / (1e-3 * N))
            assert lib.delete_filter_output(fo) == 0
        assert lib.delete_filter_input(fin) == 0


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


def test_notch_matches_oracle(hip):
    """filter.c:707-727 (experimental, no caller in the reference): kh_notch_block over a block against the oracle's
    sample-by-sample restatement, the state handed from call to call, and the per-sample entry point notch()."""
    import ctypes as C
    f, bw = 0.0123, 0.01
    rng = np.random.default_rng(5)
    n = 4000
    t = np.arange(n)
    x = (0.5 * np.exp(2j * np.pi * f * t) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    nf = hip.lib().notch_create(f, bw)
    assert nf
    ref = x.copy()

    class ONF(C.Structure):  # struct orc_notchfilter, oracle/ka9q_oracle.h
        _fields_ = [("phase", C.c_double * 2), ("step", C.c_double * 2), ("dc", C.c_float * 2), ("bw", C.c_float)]
    st = ONF()
    orc.lib().orc_notch_init(C.byref(st), f, bw)
    orc.lib().orc_notch_block(C.byref(st), ref.ctypes.data_as(C.c_void_p), n)
    got = x.copy()
    for a, b in ((0, 1500), (1500, 1501), (1501, n)):  # three calls: the state carries
        seg = np.ascontiguousarray(got[a:b])
        assert hip.lib().kh_notch_block(nf, seg.ctypes.data, b - a) == 0, hip.last_error()
        got[a:b] = seg
    assert rel_rms(got, ref) < 1e-6
    # the notch works: the tone is gone once the one-pole tracker has settled (time constant 1/bw samples)
    assert np.abs(got[2000:]).std() < 0.12 and np.abs(x[2000:]).mean() > 0.45
    assert abs(nf.contents.phase_re ** 2 + nf.contents.phase_im ** 2 - 1) < 1e-9
    # per-sample entry point, as the reference declares it
    nf2 = hip.lib().notch_create(f, bw)
    y0 = hip.lib().notch(nf2, hip.CF32(float(x[0].real), float(x[0].imag)))
    assert abs(complex(y0.re, y0.im) - complex(ref[0])) < 1e-6
    bad = hip.lib().notch(None, hip.CF32(1.0, 0.0))
    assert np.isnan(bad.re)
    C.CDLL(None).free(C.cast(nf, C.c_void_p))   # notch_delete(x) is free(x), filter.h:107
    C.CDLL(None).free(C.cast(nf2, C.c_void_p))


def test_filter_input_file_maps_the_spectrum(hip, tmp_path):
    """create_filter_input_file (filter.c:96-160): fdomain[] is a shared mapping of the named file, so every block's
    spectrum can be read from the file by another process; otherwise it behaves as create_filter_input."""
    L, M, D = 3840, 4353, 4
    N = L + M - 1
    path = str(tmp_path / "spectrum.cf32")
    lib = hip.lib()
    fin = lib.create_filter_input_file(L, M, orc.COMPLEX, path.encode())
    assert fin, hip.last_error()
    assert fin.contents.fd >= 0
    o = orc.Filter(L, M, D)
    x = _noise(2 * L, 21)
    inp = hip._as_np(fin.contents.input.c, L, np.complex64)
    for b in range(2):
        inp[:] = x[b * L:(b + 1) * L]
        assert lib.execute_filter_input(fin) == 0, hip.last_error()
        o.execute_input(x[b * L:(b + 1) * L])
        on_disk = np.fromfile(path, np.complex64)  # through the page cache: the mapping is MAP_SHARED
        assert on_disk.size == N
        assert rel_rms(on_disk, o.fdomain()) < TOL, b
    assert lib.delete_filter_input(fin) == 0
    assert not lib.create_filter_input_file(L, M, orc.COMPLEX, None)
    o.close()
