#!/usr/bin/env python3
"""bench.py -- throughput of the receive DSP hot path on MI355X (BASELINE.json metric).

One "step" = one kh_bank_execute over T blocks of the configs[1] workload (8 MS/s complex float wideband,
N = 2^18 overlap-save blocks, 64 NBFM 12.5 kHz channels, D = 256 -> 31.25 kS/s PCM): ingest (convert + LO2
mix + IF power), forward FFT, compute_n0, 64 x (bin select x response x IFFT), 64 FM demodulators, scaleclip.
The raw I/Q of the step is resident in HBM before the timed region; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Before the W warm-up steps a fixed 0.25 s of untimed steps lets the GPU leave its idle clocks (KA9Q_BENCH_PREWARM_S
overrides it; the counter passes of tools/profile_set.sh set 0); the K timed steps and the barriers are as the contract says.

N > 1 is launched by torch.distributed.run, one rank per GPU: channels shard disjointly, every GPU recomputes
the forward FFT of the same stream (SURVEY.md 8(e) option A); there is no data-path collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))

FS, L, M, D = 8_000_000, 131072, 131073, 256
N = L + M - 1
ND = N // D
OLEN = L // D
C = 64
T = 32
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def channel_plan(rank):
    """64 channels per GPU on a 12.5 kHz raster; rank r owns k = -32+64r .. 31+64r (contiguous in frequency)."""
    import synth
    ks = [k + 64 * rank for k in range(-32, 32)]
    plan = []
    for j, k in enumerate(ks):
        amp = 0.01 * (10.0 ** (j / 63.0))  # 0.01 .. 0.1 log spaced
        plan.append((k, amp, 300.0 + 50.0 * (j % 64), 2500.0))
    return plan, [synth.rotate_for(k, 12500.0, N, FS) for k in ks]


def make_input_torch(plan, device):
    """noise + one NBFM carrier per channel, T*L samples, generated on the GPU (setup, not timed)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(0x6B613971)
    n = T * L
    x = torch.randn(n, 2, device=device, generator=g, dtype=torch.float32) * 0.01
    t = torch.arange(n, device=device, dtype=torch.float64) / FS
    import synth
    for (k, amp, tone, dev) in plan:
        fc = synth.rotate_for(k, 12500.0, N, FS) * FS / N
        ph = 2 * np.pi * fc * t + (dev / tone) * torch.sin(2 * np.pi * tone * t) + 0.1 * k
        ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
        x[:, 0] += amp * torch.cos(ph)
        x[:, 1] += amp * torch.sin(ph)
    return x.contiguous()


def stage_bytes(bincount, n0_rides):
    """algorithmic bytes per launch of each stage (DESIGN.md 'Kernels'): what the kernel must read + write.
    fft1 = kh_fft_cols reading the raw ring itself (8 B/sample cf32, N samples per block window) and writing the
    column-transformed block; chan = kh_back_a: per (channel, block) N_dec spectrum bins + N_dec response + the FM work
    record per output sample, plus compute_n0's second iteration (8 B per counted bin) when it rides along."""
    return {
        "ingest": T * L * (8 + 8),
        "fft1": T * 16 * N,
        "fft2": T * 16 * N,
        "n0": T * 8 * bincount,
        "chan": C * T * (8 * ND + 8 * ND + 8 * OLEN) + (T * 8 * bincount if n0_rides else 0),
        "demod": C * T * (8 * OLEN + 2 * OLEN),
        "carry": 2 * 8 * (M - 1),
    }


def cpu_baseline(x_host, rots, budget_s=12.0):
    """the oracle (kind 'port': the CPU restatement, reference COPTS) on the same workload, single thread, bounded
    sample.  TEST INFRASTRUCTURE used as a reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pipeline

    def run(threads, budget):
        pipe = pipeline.Pipeline(L, M, D, FS, compute_n0=True, n0_low=-5500.0, n0_high=5500.0, fast=True, threads=threads)
        for r in rots:
            pipe.add_channel(demod="fm", rotate=r, low=-5500.0, high=5500.0, kaiser_beta=3.0)
        nb_avail = x_host.size // L
        t0 = time.perf_counter()
        nb = 0
        while True:  # the sample repeats circularly; the NCO/demodulator state simply runs on
            k = nb % nb_avail
            pipe.run_block([x_host[k * L:(k + 1) * L]])
            nb += 1
            if time.perf_counter() - t0 > budget:
                break
        return nb, time.perf_counter() - t0

    # One thread.  (pipeline.Pipeline(threads=n) partitions the channels over n threads behind one shared forward FFT, as
    # SURVEY.md 8(d) sketches; it gains under 10 % here because the port's forward transform, mix and compute_n0 are
    # four fifths of a block and run on one thread -- so the single-thread figure is the one reported.)
    nb1, dt1 = run(1, budget_s)
    return {"value": nb1 * L / dt1 / 1e6, "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"{nb1} blocks of {L} samples, {C} NBFM channels, {dt1:.1f} s of CPU time",
            "ms_per_block": 1e3 * dt1 / nb1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--blocks", type=int, default=T, help="blocks per step (per kh_bank_execute call)")
    ap.add_argument("--demod", default="fm", choices=["fm", "none"], help="diagnostic: 'none' runs the filter bank without demodulators")
    args = ap.parse_args()
    globals()["T"] = args.blocks

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box (not a measurement): KA9Q_BENCH_REHEARSE=1 puts every rank on device 0 and runs the
    # collectives over gloo on host tensors, so that the N > 1 control flow can be exercised without N GPUs.
    rehearse = os.environ.get("KA9Q_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else device  # where collective operands live
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import pyka9q
    assert pyka9q.lib().ka9q_hip_init(local_rank) == 0, pyka9q.last_error()

    plan, rots = channel_plan(rank)
    x = make_input_torch(plan, device)
    torch.cuda.synchronize()

    bank = pyka9q.Bank(L, M, D, 1, C, T, FS, payload=pyka9q.KH_PT_CF32, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
    bank.keep_audio(0)
    bank.keep_baseband(0)  # FM: filter output -> discriminator inside one kernel; PCM + status are the outputs
    for c, r in enumerate(rots):
        bank.set_channel(c, demod=pyka9q.KH_DEMOD_FM if args.demod == "fm" else pyka9q.KH_DEMOD_NONE, rotate=r,
                         low=-5500.0, high=5500.0, kaiser_beta=3.0)
    bank.write_input_device(0, x.data_ptr(), T * L)  # device -> device; the ring now holds the step's samples

    def step():
        bank.commit_input(0, T * L)
        bank.execute(T, sync=False)

    def barrier():
        bank.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: a step is ~70 us, so a short --warmup ends before the GPU has left its idle clocks (measured: 80 us per
    # step for 5 + 30 steps from cold against 70 us once warm).  A fixed 0.25 s of untimed steps comes first, whatever W is.
    PREWARM_S = float(os.environ.get("KA9Q_BENCH_PREWARM_S", "0.25"))  # (0 for the counter passes of tools/profile_set.sh)
    tp = time.perf_counter()
    while time.perf_counter() - tp < PREWARM_S:
        for _ in range(50):
            step()
        bank.sync()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel durations: the same K steps with a HIP event pair around every stage on the bank's stream
    bank.profile(1)
    for _ in range(args.steps):
        step()
    bank.sync()
    st = bank.stage_times()
    bank.profile(0)

    # SURVEY.md 8(e) / BASELINE configs[3]: every GPU recomputes the forward FFT of the shared stream (option A, what the
    # timed region does) against an RCCL broadcast over xGMI of the step's raw input (option B: saves host fan-out, not
    # the FFT) or of its spectrum (option C: saves the FFT on the peers).  Timed here, outside the timed region.
    probe = None
    if dist is not None:
        try:
            def bcast_ms(t):
                for _ in range(3):
                    dist.broadcast(t, src=0)
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                tb = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    dist.broadcast(t, src=0)
                torch.cuda.synchronize()
                v = torch.tensor([(time.perf_counter() - tb) / reps], device=cdev, dtype=torch.float64)
                dist.all_reduce(v, op=dist.ReduceOp.MAX)
                return round(1e3 * float(v.item()), 4)
            raw_like = (x if rank == 0 else torch.empty_like(x)).to(cdev)
            spec_like = torch.empty(T * N, 2, device=cdev, dtype=torch.float32)
            probe = {"broadcast_input_ms": bcast_ms(raw_like), "input_bytes": int(raw_like.numel() * 4),
                     "broadcast_spectrum_ms": bcast_ms(spec_like), "spectrum_bytes": int(spec_like.numel() * 4)}
            del spec_like
        except Exception as e:  # the probe must never cost the bench line
            probe = {"error": repr(e)}

    # sanity: the run demodulated something (squelch open on the strongest channel)
    stat = bank.status(C - 1, T)
    pcm = bank.pcm(C - 1, T)
    assert args.demod != "fm" or (stat["squelch_open"].max() > 0 and np.abs(pcm).max() > 100), "bench produced no audio"

    if rank == 0:
        bincount = N - (int((N * 5500.0) / FS) - int((N * -5500.0) / FS))
        # compute_n0's second iteration shares kh_back_a's launch for this shape (kh_bank.cpp enqueue_back)
        n0_rides = (ND == 1024 and 2 * OLEN == ND and os.environ.get("KA9Q_HIP_RIDE_N0", "1") != "0"
                    and os.environ.get("KA9Q_HIP_CHAN_WAVE", "1") != "0")
        sb = stage_bytes(bincount, n0_rides)
        stages = {}
        # stages whose work rides in another kernel still record an (empty) event pair of ~5 us: no rate for those
        fused_ingest = N > 8192 and os.environ.get("KA9Q_HIP_FUSED_INGEST", "1") != "0"
        empty = {"ingest": "in fft1 (kh_fft_cols reads the raw ring)"} if fused_ingest else {}
        if n0_rides:
            empty["n0"] = "iteration 1 in chan (kh_back_a), final sums in demod (kh_fm_finish_w)"
        for name, (ms, n) in st.items():
            if n:
                avg = ms / n
                if name in empty:
                    stages[name] = {"ms": round(avg, 5), "GBps": None, "empty": empty[name]}
                else:
                    stages[name] = {"ms": round(avg, 5), "GBps": round(sb[name] / (avg * 1e-3) / 1e9, 1) if avg > 0 else None}
        dom = max((k for k in stages if k != "carry" and k not in empty), key=lambda k: stages[k]["ms"])
        traffic = None  # HBM bytes per launch of that kernel from the committed PMC passes (profiles/, tools/summarize_prof.py)
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_step8_hbm_traffic.json")))["kernels"]
            key = {"fft1": "kh_fft_cols", "fft2": "kh_fft_rows", "chan": "kh_back_a", "demod": "kh_fm_finish_w", "n0": "kh_n0_pass1"}.get(dom)
            hit = [v for k, v in prof.items() if key and key in k]
            if hit and world == 1 and T == 32:
                traffic = hit[0]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        dom_ms = stages[dom]["ms"]
        achieved = sb[dom] / (dom_ms * 1e-3) / 1e9
        kernel_name = pyka9q.lib().kh_stage_kernel_name(pyka9q.KH_STAGES.index(dom)).decode()
        total_alg = T * (16 * N + C * (16 * ND + 2 * OLEN))  # BASELINE.md section 3 formula
        ms_per_step = 1e3 * dt / args.steps
        out = {
            "metric": "aggregate MS/s (input samples consumed per second, all GPUs) on 8 MS/s wideband, 64 NBFM channels per GPU",
            "value": round(world * args.steps * T * L / dt / 1e6, 2),
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_s": PREWARM_S,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)",
            "config": {"workload": "configs[1]: 8 MS/s cf32 wideband, L=131072 M=131073 N=262144, D=256 -> 31.25 kS/s, "
                                   "64 NBFM 12.5 kHz channels per GPU, 32 blocks (4.19 MS, 0.524 s of signal) per step",
                       "blocks_per_step": T, "channels_per_gpu": C, "sharding": "channels disjoint per GPU, FFT recomputed per GPU"},
            "channel_MS_per_s": round(world * C * args.steps * T * L / dt / 1e6, 1),
            "realtime_factor": round((T * L / FS) / (dt / args.steps), 1),
            "pipeline_GBps_algorithmic": round(total_alg / (ms_per_step * 1e-3) / 1e9, 1),
            "stages": stages,
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": "profiles/r01_step8_hbm_traffic.json (separate rocprofv3 --pmc passes of this command)" if traffic else None,
                         "algorithmic_bytes_per_launch": sb[dom], "avg_launch_ms": dom_ms},
        }
        if probe is not None:
            fft_ms = sum(stages[k]["ms"] for k in ("ingest", "fft1", "fft2") if k in stages)
            probe["recompute_forward_fft_ms"] = round(fft_ms, 4)
            if "broadcast_spectrum_ms" in probe:
                probe["cheaper"] = "recompute per GPU" if fft_ms <= probe["broadcast_spectrum_ms"] else "spectrum broadcast"
            out["exchange_probe"] = probe
        if not args.no_cpu:
            nb = 6
            out["cpu_baseline"] = cpu_baseline(x[:nb * L].cpu().numpy().view(np.complex64).reshape(-1), rots)
        print(json.dumps(out), flush=True)
    bank.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
