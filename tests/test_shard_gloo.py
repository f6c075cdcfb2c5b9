"""Multi-GPU path on CPU: world_size-2 `gloo` run of the channel sharding (ka9q-radio_amd/shard.py).

Each rank receives the same input blocks (input fan-out, SURVEY.md 8(e) option A), recomputes the forward FFT and
demodulates only its own contiguous-in-frequency group of channels -- with the CPU oracle standing in for the GPU
bank, since this test has no GPU -- and rank 0 gathers the PCM.  The gathered result must equal the unsharded run
bit for bit: channels are independent after the shared forward transform, so sharding changes nothing."""
import os
import socket
import sys

import numpy as np
import pytest

import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS, L, M, D = 192_000, 3840, 4353, 4
N = L + M - 1
NBLOCKS = 3
KS = [-3, -1, 0, 2, 5, 7, 6]  # 12.5 kHz raster positions, deliberately unsorted and odd in number


def _workload():
    import synth
    sigs = [(k, 0.05 + 0.01 * i, 300.0 + 100 * i, 2500.0) for i, k in enumerate(KS)]
    x = synth.wideband_fm(0, NBLOCKS * L, FS, N, sigs)
    rots = [synth.rotate_for(k, 12500.0, N, FS) for k in KS]
    return x, rots


def _run_channels(x, rots):
    import orc
    import pipeline
    pipe = pipeline.Pipeline(L, M, D, FS)
    for r in rots:
        pipe.add_channel(demod="fm", rotate=r, low=-5500.0, high=5500.0, kaiser_beta=3.0)
    olen = L // D
    pcm = np.zeros((len(rots), NBLOCKS * olen), np.int16)
    for b in range(NBLOCKS):
        outs, _ = pipe.run_block([x[b * L:(b + 1) * L]])
        for c, (_y, audio, _sig) in enumerate(outs):
            pcm[c, b * olen:(b + 1) * olen] = orc.scaleclip(audio)
    return pcm


def _worker(rank, world, port, outfile):
    for p in (os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, rots = _workload()
        plan = shard.ShardPlan(rots, N, world, rank)
        local = _run_channels(x, plan.local_rotates())
        dist.barrier()
        full = plan.gather(local, dist)
        if rank == 0:
            np.save(outfile, full)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_is_disjoint_contiguous_and_balanced():
    rots = [-4096, 4000, 0, 2, -2, 100, 8190, 8000, 50]  # 8190 == -2 mod 8192, 8000 == -192
    for world in (1, 2, 3, 4, 8, 16):
        groups = shard.partition(rots, 8192, world)
        allidx = np.concatenate(groups)
        assert sorted(allidx.tolist()) == list(range(len(rots)))
        sizes = [len(g) for g in groups]
        assert max(sizes) - min(sizes) <= 1
        prev_hi = -10 ** 9
        for g in groups:
            if len(g) == 0:
                continue
            s = [shard.signed_bin(rots[i], 8192) for i in g]
            assert min(s) >= prev_hi  # contiguous in frequency across ranks
            prev_hi = max(s)
    assert shard.signed_bin(8190, 8192) == -2 and shard.signed_bin(4096, 8192) == 4096 and shard.signed_bin(-4096, 8192) == 4096
    assert shard.bin_span([10, -10], 8192, 2048) == (-10 - 1023, 10 + 1024)


def test_two_rank_gloo_shards_equal_unsharded(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    x, rots = _workload()
    want = _run_channels(x, rots)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert np.abs(want).max() > 1000  # the squelch opened: this compared audio, not zeros
