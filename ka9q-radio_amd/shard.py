"""Multi-GPU layout of a channel bank: one process per GPU, channels sharded disjointly and
contiguously in frequency, no collective on the data path.

The reference has no in-process multi-channel receiver: it runs N `radio` processes that all listen to the same
multicast I/Q stream (README.md:383-390).  This module is that fan-out for one node: every rank receives the same
input block (host fan-out, or an RCCL broadcast of the raw block when only one rank ingests) and recomputes the
forward FFT, then serves only its own channels.  The only collectives are control-plane ones: the barrier and the
gather of results that tests and bench.py use.

torch.distributed is plumbing here (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
import os

import numpy as np


def signed_bin(rotate, N):
    """execute_filter_output's `rotate` (filter.c:240) as a signed bin offset in (-N/2, N/2]."""
    r = int(rotate) % N
    return r - N if r > N // 2 else r


def partition(rotates, N, world_size):
    """Split channels into `world_size` groups, contiguous in frequency and equal in size to within one channel.
    Returns a list (one entry per rank) of arrays of global channel indices, each sorted by frequency."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = np.argsort([signed_bin(r, N) for r in rotates], kind="stable")
    C = len(order)
    base, extra = divmod(C, world_size)
    out, pos = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(np.sort(order[pos:pos + n]) if n else np.empty(0, np.int64))
        pos += n
    return out


def bin_span(rotates, N, ND):
    """[lo, hi] signed wideband bins a rank needs for its channels (each channel reads ND bins around its
    rotate, filter.c:274-289): what a sliced spectrum exchange (SURVEY.md 8(e) option C) would have to send."""
    if len(rotates) == 0:
        return None
    s = [signed_bin(r, N) for r in rotates]
    return min(s) - (ND // 2 - 1), max(s) + ND // 2


def env_rank():
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class ShardPlan:
    """Which rank serves which channel, and how per-channel results travel back to rank 0."""

    def __init__(self, rotates, N, world_size, rank):
        self.rotates = list(rotates)
        self.N = N
        self.world_size = world_size
        self.rank = rank
        self.groups = partition(self.rotates, N, world_size)
        self.local = self.groups[rank]
        owner = np.full(len(self.rotates), -1, np.int64)
        for r, g in enumerate(self.groups):
            owner[g] = r
        assert (owner >= 0).all()
        self.owner = owner

    def local_rotates(self):
        return [self.rotates[i] for i in self.local]

    def gather(self, local_rows, dist=None, device=None):
        """local_rows: array [len(self.local), K] (e.g. PCM words of one launch).  Returns on rank 0 the array
        [C, K] in global channel order, elsewhere None.  Ranks may hold different channel counts."""
        local_rows = np.ascontiguousarray(local_rows)
        if self.world_size == 1 or dist is None:
            out = np.empty((len(self.rotates),) + local_rows.shape[1:], local_rows.dtype)
            out[self.local] = local_rows
            return out
        import torch
        K = local_rows.shape[1:]
        maxn = max(len(g) for g in self.groups)
        pad = np.zeros((maxn,) + K, local_rows.dtype)
        pad[:len(self.local)] = local_rows
        t = torch.from_numpy(pad.view(np.uint8).reshape(maxn, -1))  # bytes: every backend moves uint8
        if device is not None:
            t = t.to(device)
        bufs = [torch.empty_like(t) for _ in range(self.world_size)] if self.rank == 0 else None
        dist.gather(t, bufs, dst=0)
        if self.rank != 0:
            return None
        out = np.empty((len(self.rotates),) + K, local_rows.dtype)
        for r, g in enumerate(self.groups):
            rows = bufs[r][:len(g)].cpu().numpy()
            out[g] = np.ascontiguousarray(rows).view(local_rows.dtype).reshape((len(g),) + K)
        return out
