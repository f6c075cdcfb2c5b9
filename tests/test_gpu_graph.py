"""hipGraph replay of the bank's two kernel sequences (kh_bank_use_graph / KA9Q_HIP_GRAPH; north_star: "hipGraph-captured
per block").  The library captures a launch shape the second time it sees it and replays the graph from the third; the
shape includes the block count, the buffer parities and -- in stream mode -- which of the two device raw rings the front
half reads.  Everything a replayed launch leaves (PCM, per-channel status, input status) must be bit-identical to plain
launches of the same samples: the kernels and their arguments are the same, only the way they are queued differs."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

FS, L, M, D, C, T = 8_000_000, 131072, 131073, 256, 6, 3
N = L + M - 1
LAUNCHES = [2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 2, 2, 1, 1, 1, 1, 1, 1]  # the block count changes in the middle, twice


def _banks(hip, n, payload, mixed=False):
    """mixed: the last two channels are USB with hang AGC and envelope AM -- a bank whose linear demodulators run on a
    stream of their own beside the FM demodulator (a fork and a join inside the captured back half)"""
    sigs = [(k - 3, 0.03 + 0.01 * k, 300.0 + 60 * k, 2500.0) for k in range(C)]
    out_rate = FS / D
    out = []
    for _ in range(n):
        b = hip.Bank(L, M, D, 1, C, T, FS, payload=payload, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
        for c, (k, *_r) in enumerate(sigs):
            rot = synth.rotate_for(k, 12500.0, N, FS)
            if mixed and c >= C - 2:
                b.set_channel(c, demod=hip.KH_DEMOD_LINEAR, rotate=rot, low=100.0 if c == C - 2 else -5000.0, high=3000.0 if c == C - 2 else 5000.0,
                              kaiser_beta=3.0, pll=0, agc=1, env=int(c == C - 1), hangtime=1.1 * out_rate,
                              recovery_rate=10 ** ((6.0 / out_rate) / 20), agc_gain=1e3)
            else:
                b.set_channel(c, demod=hip.KH_DEMOD_FM, rotate=rot, low=-5500.0, high=5500.0, kaiser_beta=3.0)
        out.append(b)
    return sigs, out


def _raw(hip, sigs):
    x = synth.wideband_fm(11, sum(LAUNCHES) * L, FS, N, sigs)
    raw = np.empty(2 * x.size, np.int16)  # IQ_PT: interleaved int16, radio.c:159-167
    raw[0::2] = np.clip(np.rint(x.real * 32767 * 4), -32768, 32767)
    raw[1::2] = np.clip(np.rint(x.imag * 32767 * 4), -32768, 32767)
    return raw


def _snapshot(b, nb):
    return (b.pcm_all(nb).copy(), b.input_status(0, nb).tobytes(), [b.status(c, nb).tobytes() for c in range(C)])


@pytest.mark.parametrize("mixed", [False, True])
def test_graph_replay_matches_plain_launches_write_input(hip, mixed):
    sigs, (plain, graph) = _banks(hip, 2, hip.KH_PT_IQ16LE, mixed)
    raw = _raw(hip, sigs)
    graph.use_graph(1)
    pos = 0
    for j, nb in enumerate(LAUNCHES):
        seg = raw[2 * pos:2 * (pos + nb * L)]
        outs = []
        for b in (plain, graph):
            b.write_input(0, seg, count=nb * L)
            b.execute(nb)
            outs.append(_snapshot(b, nb))
        assert np.array_equal(outs[0][0], outs[1][0]), j
        assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2], j
        pos += nb * L
    assert np.abs(outs[1][0]).max() > 100
    # switching the graphs off again drops them; the next launch is a plain one and still agrees
    graph.use_graph(0)
    for b in (plain, graph):
        b.write_input(0, raw[:2 * L], count=L)
        b.execute(1)
    assert np.array_equal(plain.pcm_all(1), graph.pcm_all(1))
    for b in (plain, graph):
        b.close()


def test_graph_replay_matches_plain_launches_stream(hip):
    """stream mode alternates the front half between two device raw rings, and the ring is part of what a captured front
    half has baked in: close and reopen after an ODD number of submits, so that ring parity and submit parity part ways"""
    sigs, (plain, graph) = _banks(hip, 2, hip.KH_PT_IQ16LE)
    raw = _raw(hip, sigs)
    graph.use_graph(1)
    pos = 0
    plain.stream_open(3)
    graph.stream_open(3)
    for j, nb in enumerate(LAUNCHES):
        if j == 9:  # nine submits so far
            for b in (plain, graph):
                b.stream_close()
                b.stream_open(2)
        outs = []
        for b in (plain, graph):
            slot = b.stream_slot(0, np.int16, 2 * nb * L)
            slot[:] = raw[2 * pos:2 * (pos + nb * L)]
            b.stream_submit(nb)
            outs.append(_snapshot(b, nb))
        assert np.array_equal(outs[0][0], outs[1][0]), j
        assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2], j
        pos += nb * L
    assert np.abs(outs[1][0]).max() > 100
    for b in (plain, graph):
        b.stream_close()
        b.close()


def test_mixed_bank_demodulators_side_by_side_or_in_sequence(hip, monkeypatch):
    """a mixed bank runs its linear / AM demodulators on a third stream beside the FM demodulator (KA9Q_HIP_DEMOD_FORK=0:
    one after the other on the back stream): the same bits either way, launch after launch without a host sync between"""
    raw = None
    snaps = []
    for fork in ("1", "0"):
        monkeypatch.setenv("KA9Q_HIP_DEMOD_FORK", fork)  # read by kh_bank_create
        sigs, (b,) = _banks(hip, 1, hip.KH_PT_IQ16LE, mixed=True)
        if raw is None:
            raw = _raw(hip, sigs)
        pos, out = 0, []
        for nb in LAUNCHES[:10]:
            b.write_input(0, raw[2 * pos:2 * (pos + nb * L)], count=nb * L)
            b.execute(nb, sync=False)
            out.append(_snapshot(b, nb))
            pos += nb * L
        b.close()
        snaps.append(out)
    for j, (a, c) in enumerate(zip(*snaps)):
        assert np.array_equal(a[0], c[0]) and a[1] == c[1] and a[2] == c[2], j
    assert np.abs(snaps[0][-1][0][C - 2:]).max() > 100  # the linear channels produce audio


@pytest.mark.parametrize("mixed", [False, True])
def test_back_half_queued_by_the_banks_own_thread(hip, monkeypatch, mixed):
    """KA9Q_HIP_ENQ_THREAD=1: kh_bank_execute queues the front half and returns; a thread of the bank's own queues the
    hand-over, the back half and the slot's event.  Same kernels, same arguments, same order on every stream: bit-identical
    to launches queued by the caller -- several launches in flight (no host sync in between), through the streaming ring as
    well, with a channel retuned in the middle (kh_bank_set_channel has to wait for the thread) and the mode's switches
    (keep_baseband, graphs on and off again) flipped while it is running."""
    banks = []
    for env in ("0", "1"):
        monkeypatch.setenv("KA9Q_HIP_ENQ_THREAD", env)  # read by kh_bank_create
        sigs, (b,) = _banks(hip, 1, hip.KH_PT_IQ16LE, mixed)
        banks.append(b)
    raw = _raw(hip, sigs)
    pos = 0
    for j, nb in enumerate(LAUNCHES):
        seg = raw[2 * pos:2 * (pos + nb * L)]
        outs = []
        for b in banks:
            if j == 5:
                b.set_channel(0, demod=hip.KH_DEMOD_FM, rotate=synth.rotate_for(-2, 12500.0, N, FS), low=-5500.0, high=5500.0, kaiser_beta=3.0)
            if j == 8:
                b.keep_baseband(0)
            if j == 12:
                b.use_graph(1)
            if j == 17:
                b.use_graph(0)
            b.write_input(0, seg, count=nb * L)
            b.execute(nb, sync=False)
            if j % 4 == 3 or j == len(LAUNCHES) - 1:
                outs.append(_snapshot(b, nb))
        if outs:
            assert np.array_equal(outs[0][0], outs[1][0]), j
            assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2], j
        pos += nb * L
    assert np.abs(outs[1][0]).max() > 100
    # the streaming ring on the same banks
    for b in banks:
        b.stream_open(3)
    pos = 0
    for j, nb in enumerate(LAUNCHES[:9]):
        outs = []
        for b in banks:
            slot = b.stream_slot(0, np.int16, 2 * nb * L)
            slot[:] = raw[2 * pos:2 * (pos + nb * L)]
            b.stream_submit(nb)
            if j % 3 == 2:
                outs.append(_snapshot(b, nb))
        if outs:
            assert np.array_equal(outs[0][0], outs[1][0]), j
            assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2], j
        pos += nb * L
    for b in banks:
        b.stream_close()
        b.close()
