"""SURVEY.md 8(e) for the mixed-mode plan (BASELINE.json configs[2]: NBFM / envelope AM / USB quarters on the 12.5 kHz
raster): every rank takes its channels from workloads.shard_plan (shard.partition: contiguous in frequency, equal counts to
within one) exactly as bench.py's N > 1 path and a multi-GPU host do.  No data-path collective exists: the ranks exchange
nothing but the results gathered here for the check.

* the per-rank parameter sets (what kh_bank_set_channel receives) of the full 1024-channel plan, gathered over `gloo` from
  two ranks, are the unsharded plan's, channel for channel;
* the same rule on a shape the CPU oracle finishes in seconds: two ranks demodulate their shares of a 12-channel mixed plan
  (the oracle standing in for the bank: this test has no GPU) and the gathered PCM equals the unsharded run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

import shard
import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = W.Shape("config 3's rule at 192 kS/s", 192_000, 3840, 4353, 4)
NSMALL, NBLOCKS = 12, 3


def _run_oracle(plan, x, shape):
    import orc
    import pipeline
    pipe = pipeline.Pipeline(shape.L, shape.M, shape.D, shape.fs)
    for ch in plan:
        pipe.add_channel(demod=ch["demod"], **W.channel_kwargs(ch))
    pcm = np.zeros((len(plan), NBLOCKS * shape.olen), np.int16)
    for b in range(NBLOCKS):
        outs, _ = pipe.run_block([x[b * shape.L:(b + 1) * shape.L]])
        for c, (_y, audio, _sig) in enumerate(outs):
            pcm[c, b * shape.olen:(b + 1) * shape.olen] = orc.scaleclip(audio)[:shape.olen]
    return pcm


def _worker(rank, world, port, outdir):
    for p in (os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) the full plan: what this rank would hand to kh_bank_set_channel
        mine, idx = W.shard_plan(W.config3_plan(), W.CONFIG3, world, rank)
        params = [(i, ch["demod"], W.channel_kwargs(ch)) for i, ch in zip(idx, mine)]
        gathered = [None] * world
        dist.all_gather_object(gathered, params)
        # (2) the small plan through the oracle
        small = W.mixed_plan(SMALL, NSMALL)
        x = W.plan_signal(small, SMALL, 0, NBLOCKS * SMALL.L)
        sp = shard.ShardPlan([ch["rotate"] for ch in small], SMALL.N, world, rank)
        local = _run_oracle([small[i] for i in sp.local], x, SMALL)
        dist.barrier()
        full = sp.gather(local, dist)
        if rank == 0:
            np.save(os.path.join(outdir, "pcm.npy"), full)
            import json
            json.dump(gathered, open(os.path.join(outdir, "params.json"), "w"))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_config3_shares_cover_the_plan(world):
    plan = W.config3_plan()
    assert len(plan) == 1024
    seen = {}
    prev_hi = -10 ** 9
    for rank in range(world):
        mine, idx = W.shard_plan(plan, W.CONFIG3, world, rank)
        assert len(mine) in (1024 // world, 1024 // world + 1)
        kinds = {k: sum(ch["kind"] == k for ch in mine) for k in ("fm", "am", "usb")}
        # the k mod 4 rule survives a contiguous cut: half NBFM, a quarter each of AM and USB, to within the cut's edges
        assert abs(kinds["fm"] - len(mine) / 2) <= 2 and abs(kinds["am"] - len(mine) / 4) <= 1 and abs(kinds["usb"] - len(mine) / 4) <= 1
        bins = [shard.signed_bin(ch["rotate"], W.CONFIG3.N) for ch in mine]
        assert min(bins) > prev_hi  # contiguous in frequency, rank after rank
        prev_hi = max(bins)
        for i, ch in zip(idx, mine):
            assert i not in seen
            seen[i] = ch
    assert sorted(seen) == list(range(1024))
    assert all(seen[i] is plan[i] for i in range(1024))


def test_two_rank_gloo_mixed_plan(tmp_path):
    import json
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    # (1) parameter sets
    plan = W.config3_plan()
    gathered = json.load(open(tmp_path / "params.json"))
    assert len(gathered) == world
    got = {}
    for rank_params in gathered:
        assert len(rank_params) == 512
        for i, demod, kw in rank_params:
            assert i not in got
            got[i] = (demod, kw)
    assert sorted(got) == list(range(1024))
    for i, ch in enumerate(plan):
        assert got[i][0] == ch["demod"]
        assert got[i][1] == json.loads(json.dumps(W.channel_kwargs(ch))), i
    # rank 0 holds the lower half of the band
    assert max(shard.signed_bin(plan[i]["rotate"], W.CONFIG3.N) for i, *_ in gathered[0]) < \
        min(shard.signed_bin(plan[i]["rotate"], W.CONFIG3.N) for i, *_ in gathered[1])
    # (2) PCM of the small mixed plan
    small = W.mixed_plan(SMALL, NSMALL)
    x = W.plan_signal(small, SMALL, 0, NBLOCKS * SMALL.L)
    want = _run_oracle(small, x, SMALL)
    pcm = np.load(tmp_path / "pcm.npy")
    assert pcm.shape == want.shape and np.array_equal(pcm, want)
    for c, ch in enumerate(small):  # every kind produced audio: this compared demodulated signals, not zeros
        assert np.abs(want[c]).max() > 100, (c, ch["kind"])
