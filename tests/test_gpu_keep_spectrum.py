"""kh_bank_keep_spectrum(bank, 0): the row pass of a four-step transform stores only the rows of the spectrum that the
channels' filter outputs read (filter.c:274-335), and compute_n0's second iteration (radio.c:451-462) never re-reads the
spectrum: a block whose bracket fails is flagged by the last row-pass workgroup of the block to finish, which saves the
block's scratch, and kh_n0_repair takes its second iteration from there.  What a launch leaves -- PCM, per-channel status,
IF power -- must be bit-identical to a bank that keeps the whole spectrum (the channel kernels read the same bins, computed
by the same instructions); n0 is bit-identical as well while the bracket holds, and the same sum in another order where
it does not, so it is held to 1e-6 and, like every route of compute_n0, to 1e-4 against a float64 evaluation of
radio.c:422-465 on the full spectrum."""
import numpy as np
import pytest

import synth
from test_gpu_configs import _n0_float64

pytestmark = pytest.mark.gpu

SHAPES = {"2^14": (1_000_000, 8192, 8193, 32), "2^18": (8_000_000, 131072, 131073, 256), "2^19": (20_000_000, 262144, 262145, 512)}


def _bank(hip, shape, ks, T, payload=None, **kw):
    fs, L, M, D = shape
    N = L + M - 1
    extra = {} if payload is None else {"payload": payload}
    b = hip.Bank(L, M, D, 1, len(ks), T, fs, compute_n0=1, n0_low=-5500.0, n0_high=5500.0, **extra)
    for c, k in enumerate(ks):
        b.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(k, 12500.0, N, fs), low=-5500.0, high=5500.0, kaiser_beta=3.0, **kw)
    return b


def _same(a, q, C, nb, what):
    assert np.array_equal(a.pcm_all(nb), q.pcm_all(nb)), what
    for c in range(C):
        assert a.status(c, nb).tobytes() == q.status(c, nb).tobytes(), (what, c)
    sa, sq = a.input_status(0, nb), q.input_status(0, nb)
    assert sa["if_power"].tobytes() == sq["if_power"].tobytes(), what
    for g, r in zip(sa["n0"], sq["n0"]):
        assert abs(float(g) - float(r)) <= 1e-6 * abs(float(r)), (what, float(g), float(r))


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("case", ["inband", "outband", "noise"])
def test_sparse_spectrum_equals_full(hip, monkeypatch, shape, case):
    """the three inputs of test_compute_n0_modes_agree: "inband" sends every block through kh_n0_repair (the bracket fails:
    three quarters of the power sit inside the excluded range; KA9Q_HIP_SPARSE_HOLD=0 keeps the bank from falling back to
    whole spectra after the first such launch), "outband" leaves no candidates, "noise" more than the slots requested in
    advance.  Channels straddle the band edge (bins wrap) and row boundaries."""
    monkeypatch.setenv("KA9Q_HIP_SPARSE_HOLD", "0")  # read by kh_bank_create
    fs, L, M, D = SHAPES[shape]
    N = L + M - 1
    T, calls = 3, 3
    sigs = {"inband": [(-2, 0.05, 400.0, 2500.0), (0, 0.2, 700.0, 2000.0), (3, 0.1, 1000.0, 3000.0)],
            "outband": [(-2, 0.05, 400.0, 2500.0), (3, 0.1, 1000.0, 3000.0)], "noise": []}[case]
    x = np.ascontiguousarray(synth.wideband_fm(11, T * calls * L, fs, N, sigs), np.complex64)
    half = int(fs / 2 / 12500.0)
    ks = [-2, 0, 3, 7, -half + 1, half - 1]
    full, sparse = _bank(hip, SHAPES[shape], ks, T), _bank(hip, SHAPES[shape], ks, T)
    sparse.keep_spectrum(0)
    prev = None
    for call in range(calls):
        seg = x[call * T * L:(call + 1) * T * L]
        for b in (full, sparse):
            b.write_input(0, seg)
            b.execute(T)
        _same(full, sparse, len(ks), T, (shape, case, call))
        for blk in range(T):
            v = _n0_float64(full.spectrum(0, blk), N, -5500.0, 5500.0, fs)
            prev = v if prev is None else prev + .005 * (v - prev)
            g = float(sparse.input_status(0, T)["n0"][blk])
            assert abs(g - prev) <= 1e-4 * prev, (shape, case, call, blk, g, prev)
        with pytest.raises(Exception):
            sparse.spectrum(0, 0)
    assert np.abs(sparse.pcm_all(T)).max() > 100 or case == "noise"
    full.close()
    sparse.close()


def test_sparse_spectrum_falls_back_while_the_bracket_fails(hip):
    """by default a bank whose blocks keep failing the bracket stores whole spectra for a while (64 launches, then twice as
    long) instead of repairing every block: the launches before the host notices, the ones in the fall-back and the ones
    after it all leave what the full bank leaves"""
    shape = SHAPES["2^18"]
    fs, L, M, D = shape
    N = L + M - 1
    T, calls = 2, 70
    sigs = [(-2, 0.05, 400.0, 2500.0), (0, 0.2, 700.0, 2000.0), (3, 0.1, 1000.0, 3000.0)]
    x = np.ascontiguousarray(synth.wideband_fm(11, 4 * T * L, fs, N, sigs), np.complex64)
    quiet = np.ascontiguousarray(synth.wideband_fm(12, 4 * T * L, fs, N, sigs[:1] + sigs[2:]), np.complex64)
    ks = [-2, 0, 3]
    full, sparse = _bank(hip, shape, ks, T), _bank(hip, shape, ks, T)
    sparse.keep_spectrum(0)
    readable = []
    for call in range(calls):
        src = x if call < 3 else quiet  # three launches with the carrier inside the excluded range, then none
        seg = src[(call % 4) * T * L:(call % 4 + 1) * T * L]
        for b in (full, sparse):
            b.write_input(0, seg)
            b.execute(T)
        _same(full, sparse, len(ks), T, call)
        try:
            sparse.spectrum(0, 0)
            readable.append(True)
        except Exception:
            readable.append(False)
    assert not readable[0]            # sparse from the start
    assert any(readable[1:8])         # the fall-back: whole spectra again
    assert not readable[-1]           # and back to the channels' rows once the hold has run out
    full.close()
    sparse.close()


def test_sparse_spectrum_follows_the_channels(hip):
    """the rows kept are those of the channels as they are tuned NOW: retune between launches, and switch the mode on and off"""
    shape = SHAPES["2^18"]
    fs, L, M, D = shape
    N = L + M - 1
    T = 2
    sigs = [(k, 0.05, 300.0 + 40 * k, 2500.0) for k in (-300, -20, 5, 130, 290)]
    x = np.ascontiguousarray(synth.wideband_fm(3, 6 * T * L, fs, N, sigs), np.complex64)
    ks = [-300, -20, 5]
    full, sparse = _bank(hip, shape, ks, T), _bank(hip, shape, ks, T)
    sparse.keep_spectrum(0)
    moves = [None, (0, 130), None, (2, 290), (1, -300), None]
    for call, mv in enumerate(moves):
        if call == 3:
            sparse.keep_spectrum(1)
        if call == 4:
            sparse.keep_spectrum(0)
        seg = x[call * T * L:(call + 1) * T * L]
        for b in (full, sparse):
            if mv is not None:
                b.set_channel(mv[0], demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(mv[1], 12500.0, N, fs), low=-5500.0, high=5500.0,
                              kaiser_beta=3.0)
            b.write_input(0, seg)
            b.execute(T, sync=False)
        _same(full, sparse, len(ks), T, call)
    full.close()
    sparse.close()


def test_sparse_spectrum_stream_and_graph(hip):
    """the streaming ring with hipGraph replay on a bank that keeps only its channels' rows: the mask is baked into the captured
    row pass, and so are the arrival counters that the column pass zeroes"""
    shape = SHAPES["2^18"]
    fs, L, M, D = shape
    N = L + M - 1
    T = 2
    ks = [-3, -2, 0, 1, 2, 200]
    sigs = [(k, 0.03, 300.0 + 60 * j, 2500.0) for j, k in enumerate(ks)]
    x = synth.wideband_fm(11, 8 * T * L, fs, N, sigs)
    raw = np.empty(2 * x.size, np.int16)
    raw[0::2] = np.clip(np.rint(x.real * 32767 * 4), -32768, 32767)
    raw[1::2] = np.clip(np.rint(x.imag * 32767 * 4), -32768, 32767)
    full, sparse = _bank(hip, shape, ks, T, payload=hip.KH_PT_IQ16LE), _bank(hip, shape, ks, T, payload=hip.KH_PT_IQ16LE)
    sparse.keep_spectrum(0)
    sparse.use_graph(1)
    for b in (full, sparse):
        b.stream_open(3)
    for call in range(8):
        for b in (full, sparse):
            slot = b.stream_slot(0, np.int16, 2 * T * L)
            slot[:] = raw[2 * call * T * L:2 * (call + 1) * T * L]
            b.stream_submit(T)
        _same(full, sparse, len(ks), T, call)
    assert np.abs(sparse.pcm_all(T)).max() > 100
    for b in (full, sparse):
        b.stream_close()
        b.close()
