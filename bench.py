#!/usr/bin/env python3
"""bench.py -- throughput of the receive DSP hot path on MI355X (BASELINE.json metric).

One "step" = one kh_bank_execute over T blocks of the configs[1] workload (8 MS/s complex float wideband,
N = 2^18 overlap-save blocks, 64 NBFM 12.5 kHz channels, D = 256 -> 31.25 kS/s PCM): ingest (convert + LO2
mix + IF power), forward FFT, compute_n0, 64 x (bin select x response x IFFT), 64 FM demodulators, scaleclip.
The raw I/Q of the step is resident in HBM before the timed region; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Before the W warm-up steps a fixed 0.25 s of untimed steps lets the GPU leave its idle clocks (KA9Q_BENCH_PREWARM_S
overrides it; the counter passes of tools/profile_set.sh set 0); the K timed steps and the barriers are as the contract says.

After the headline region (rank 0, one GPU; all of it outside the timed region and reported as extra keys of the one
JSON line): `shapes` -- the other BASELINE configurations as one GPU sees them (config 4's per-GPU share at 64 MS/s /
N = 2^20, config 3's mixed AM/FM/SSB share at 20 MS/s, config 5's 256 PLL pipelines), each with its step time, its
SURVEY.md 8(d) bytes and the fraction of the HBM roofline those amount to; `realtime` -- BASELINE metric (i): the whole
8 MS/s raster (640 channels) fed block by block in real time from the library's pinned host ring through hipMemcpyAsync
(s16le wire format), 1000 consecutive blocks, plus the same path unpaced for the PCIe-inclusive rate; `cpu_baseline` --
the oracle on the host cores, one thread and all the box's cores.

N > 1 is launched by torch.distributed.run, one rank per GPU: channels shard disjointly and contiguously in frequency
(ka9q-radio_amd/shard.py), every GPU recomputes the forward FFT of the same stream (SURVEY.md 8(e) option A); there is
no data-path collective.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import workloads  # noqa: E402

SH = workloads.CONFIG2
FS, L, M, D, N, ND, OLEN = SH.fs, SH.L, SH.M, SH.D, SH.N, SH.ND, SH.olen
C = 64
T = 32
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def make_wideband_torch(shape, plan, nsamp, device, seed=0x6B613971, max_signals=96):
    """noise (sigma 0.01 per rail) + the plan's carriers, nsamp samples at the wideband rate, generated on the GPU (setup,
    not timed).  NBFM / AM / two-tone USB as ka9q-radio_amd/synth.py defines them.  Only the first `max_signals` channels
    (evenly spread over the plan) carry a signal when the plan is larger: the arithmetic does not depend on it and the
    set-up stays short; the others demodulate noise (squelch closed / AGC at full gain)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(nsamp, 2, device=device, generator=g, dtype=torch.float32) * 0.01
    t = torch.arange(nsamp, device=device, dtype=torch.float64) / shape.fs
    step = max(1, len(plan) // max_signals)
    for ch in plan[::step]:
        fc = ch["rotate"] * shape.fs / shape.N
        if ch["kind"] == "fm":
            ph = 2 * np.pi * fc * t + (ch["deviation"] / ch["tone"]) * torch.sin(2 * np.pi * ch["tone"] * t) + 0.1 * ch["k"]
            env = ch["amp"]
        elif ch["kind"] == "am":
            ph = 2 * np.pi * fc * t + 0.3
            env = (ch["amp"] * (1.0 + ch["depth"] * torch.cos(2 * np.pi * ch["tone"] * t))).to(torch.float32)
        else:  # usb two-tone: sum of two carriers
            for f in (700.0, 1900.0):
                p2 = torch.remainder(2 * np.pi * (fc + f) * t, 2 * np.pi).to(torch.float32)
                x[:, 0] += 0.5 * ch["amp"] * torch.cos(p2)
                x[:, 1] += 0.5 * ch["amp"] * torch.sin(p2)
            continue
        ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
        x[:, 0] += env * torch.cos(ph)
        x[:, 1] += env * torch.sin(ph)
    return x.contiguous()


def make_pipelines_torch(shape, plan, nsamp, device, seed=5):
    """config 5: one 192 kS/s input per pipeline, a DSB signal with (CAM) or without (`square`) carrier at the IF, swept
    by (f0, fdot); [P, nsamp, 2] float32"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    P = len(plan)
    x = torch.randn(P, nsamp, 2, device=device, generator=g, dtype=torch.float32) * 0.001
    t = (torch.arange(nsamp, device=device, dtype=torch.float64) / shape.fs)[None, :]
    f0 = torch.tensor([ch["if_hz"] + ch["f0"] for ch in plan], device=device, dtype=torch.float64)[:, None]
    fd = torch.tensor([ch["fdot"] for ch in plan], device=device, dtype=torch.float64)[:, None]
    carrier = torch.tensor([0.0 if ch["square"] else 1.0 for ch in plan], device=device, dtype=torch.float32)[:, None]
    ph = torch.remainder(2 * np.pi * (f0 * t + 0.5 * fd * t * t), 2 * np.pi).to(torch.float32)
    env = 0.05 * (carrier + 0.5 * torch.cos(2 * np.pi * 400.0 * t).to(torch.float32))
    x[:, :, 0] += env * torch.cos(ph)
    x[:, :, 1] += env * torch.sin(ph)
    return x.contiguous()


def build_bank(pyka9q, shape, plan, nblocks, payload=None, compute_n0=1):
    ids = {"fm": pyka9q.KH_DEMOD_FM, "linear": pyka9q.KH_DEMOD_LINEAR}
    bank = pyka9q.Bank(shape.L, shape.M, shape.D, shape.P, len(plan), nblocks, shape.fs,
                       payload=pyka9q.KH_PT_CF32 if payload is None else payload, compute_n0=compute_n0, n0_low=-5500.0, n0_high=5500.0)
    bank.keep_audio(0)
    bank.keep_spectrum(KEEP_SPECTRUM)  # 0: the row pass stores only the rows of the spectrum that the channels read
    if all(ch["demod"] == "fm" for ch in plan):
        bank.keep_baseband(0)  # FM: filter output -> discriminator inside one kernel; PCM + status are the outputs
    for c, ch in enumerate(plan):
        bank.set_channel(c, demod=ids[ch["demod"]], **workloads.channel_kwargs(ch))
    return bank


# PCM, channel status and input status are a bank's outputs; the N-bin spectrum of every block is an intermediate that only
# kh_bank_read_spectrum hands out.  0 (--keep-spectrum 0, the default): four-step row passes store the rows of N1 bins that a
# channel's filter output reads and nothing else (kh_bank_keep_spectrum; bit-identical PCM and status:
# tests/test_gpu_keep_spectrum.py).  SURVEY.md 8(d)'s per-step bytes still count the whole spectrum's write.
KEEP_SPECTRUM = 0


def kept_spectrum_bins(n, nd, rotates):
    """bins per block that the row pass stores with kh_bank_keep_spectrum(0): the rows (N1 bins) that hold the N_dec bins
    around each channel's rotate, in groups of four rows (a row-pass wavefront's share: kh_fft_rows, SpecMask)"""
    if n <= 8192:
        return n
    n1 = 1 << (int(np.log2(n)) // 2)
    n2 = n // n1
    groups = set()
    for r in rotates:
        for row in range((r - nd // 2) // n1, (r + nd // 2) // n1 + 1):
            groups.add((row % n2) // 4)
    return min(n, 4 * n1 * len(groups))


def stage_bytes(bincount, n0_rides, kept_bins=None):
    """algorithmic bytes per launch of each stage of the headline shape (DESIGN.md 'Kernels'): what the kernel must read +
    write.  fft1 = kh_fft_cols reading the raw ring itself (8 B/sample cf32, N samples per block window) and writing the
    column-transformed block; chan = kh_back_a: per (channel, block) N_dec spectrum bins + N_dec response + the FM work
    record per output sample, plus compute_n0's second iteration (8 B per counted bin) when it rides along."""
    return {
        "ingest": T * L * (8 + 8),
        "fft1": T * 16 * N,
        "fft2": T * (8 * N + 8 * (N if kept_bins is None else kept_bins)),
        "n0": T * 8 * bincount,
        "chan": C * T * (8 * ND + 8 * ND + 8 * OLEN) + (T * 8 * bincount if n0_rides else 0),
        "demod": C * T * (8 * OLEN + 2 * OLEN),
        "carry": 2 * 8 * (M - 1),
    }


def time_steps(bank, step, steps, warm=5, warm_s=0.15):
    # (at least `warm` steps and warm_s seconds, so that the clocks have left their idle state: 40 steps of config 4's share
    # timed after five warm-up steps came to 118 us per step where 200 steps after a warm-up give 111)
    t0 = time.perf_counter()
    k = 0
    while k < warm or time.perf_counter() - t0 < warm_s:
        step()
        k += 1
        if k % 8 == 0:
            bank.sync()
    bank.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    bank.sync()
    return (time.perf_counter() - t0) / steps


def measure_shape(pyka9q, torch, device, shape, plan, nblocks, label, steps=40, valu_key=None):
    """one of the other BASELINE configurations on this GPU: step time (as the headline: raw input resident, launches
    back to back), per-stage HIP-event times, SURVEY.md 8(d) bytes and what fraction of the HBM roofline they are"""
    P = shape.P
    if P == 1:
        x = make_wideband_torch(shape, plan, nblocks * shape.L, device)
    else:
        x = make_pipelines_torch(shape, plan, nblocks * shape.L, device)
    torch.cuda.synchronize()
    bank = build_bank(pyka9q, shape, plan, nblocks)
    if P == 1:
        bank.write_input_device(0, x.data_ptr(), nblocks * shape.L)
    else:
        for i, ch in enumerate(plan):
            bank.write_input_device(i, x[i].data_ptr(), nblocks * shape.L)
            bank.set_lo(i, ch["lo2"], ch["doppler"], ch["doppler_rate"])

    def step():
        for i in range(P):
            bank.commit_input(i, nblocks * shape.L)
        bank.execute(nblocks, sync=False)

    dt = time_steps(bank, step, steps)
    bank.profile(1)  # every stage alone on the chip
    for _ in range(5):
        step()
    bank.sync()
    st = bank.stage_times()
    bank.profile(2)  # the two-stream pipeline, event pairs on the stream each stage runs on
    for _ in range(12):
        step()
    bank.sync()
    st2 = bank.stage_times()
    bank.profile(0)
    bank.close()
    del x
    torch.cuda.empty_cache()
    alg = shape.algorithmic_bytes(plan, nblocks)
    kinds = {}
    for ch in plan:
        kinds[ch["kind"]] = kinds.get(ch["kind"], 0) + 1
    return {"workload": label, "channels": len(plan), "kinds": kinds, "inputs": P, "N": shape.N, "blocks_per_step": nblocks,
            "us_per_step": round(dt * 1e6, 1), "MSps": round(P * nblocks * shape.L / dt / 1e6, 1),
            "realtime_factor": round((nblocks * shape.L / shape.fs) / dt, 1),
            "algorithmic_bytes_per_step": alg, "algorithmic_GBps": round(alg / dt / 1e9, 1),
            "frac_of_hbm_peak": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4),
            "stage_us_hip_events": {k: round(1e3 * ms / n, 1) for k, (ms, n) in st.items() if n},
            "stage_us_pipelined": {k: round(1e3 * ms / n, 1) for k, (ms, n) in st2.items() if n},
            "back_half": back_half_ceilings(shape, plan, nblocks, st, st2, valu_key)}


SIMDS, CLOCK_HZ = 1024, 2.4e9  # /opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock


def back_half_ceilings(shape, plan, nblocks, st_alone, st_pipe, valu_key):
    """the fused filter / demodulator kernels against BOTH of their ceilings: SURVEY.md 8(d)'s bytes per (channel, block)
    (8 N_dec bins + 8 N_dec response + b_out olen) over their HIP-event durations as a fraction of the HBM peak, and --
    they are bound by instruction issue, not by bytes -- the vector instructions one launch issues (SQ_INSTS_VALU from
    the committed rocprofv3 --pmc pass of the same shape, profiles/r03_valu_counts.json) x 4 cycles / 1024 SIMDs / 2.4 GHz
    against the same duration."""
    alg = 0
    for ch in plan:
        alg += nblocks * (16 * shape.ND + (4 if ch.get("channels", 1) == 2 else 2) * shape.olen)
    out = {"algorithmic_bytes_per_launch": alg}
    for tag, st in (("alone", st_alone), ("pipelined", st_pipe)):
        us = sum(1e3 * st[k][0] / st[k][1] for k in ("chan", "demod") if st.get(k, (0, 0))[1])
        if us > 0:
            out[f"us_{tag}"] = round(us, 1)
            out[f"frac_of_hbm_peak_{tag}"] = round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
    try:
        counts = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_counts.json")))[valu_key]
        insts = sum(k["SQ_INSTS_VALU"] for k in counts["kernels"].values())
        # a launch with fewer wavefronts than the chip has SIMDs can only issue on as many SIMDs as it has wavefronts
        # (config 5's PLL kernel: 4 wavefronts, one lane per pipeline)
        bound_us = sum(k["SQ_INSTS_VALU"] * 4 / min(SIMDS, max(1.0, k["SQ_WAVES"])) for k in counts["kernels"].values()) / CLOCK_HZ * 1e6
        out["roofline_valu"] = {"bound": "valu_issue", "SQ_INSTS_VALU_per_launch": int(insts), "kernels": sorted(counts["kernels"]),
                                "issue_bound_us": round(bound_us, 1), "source": counts["source"],
                                "frac_alone": round(bound_us / out["us_alone"], 4) if out.get("us_alone") else None,
                                "frac_pipelined": round(bound_us / out["us_pipelined"], 4) if out.get("us_pipelined") else None}
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return out


def shapes_leg(pyka9q, torch, device):
    out = []
    jobs = [
        (workloads.CONFIG4, workloads.shard_plan(workloads.config4_plan(), workloads.CONFIG4, 8, 3)[0], 16,
         "config 4, one GPU's share of 8: 64 MS/s cf32, N = 2^20, D = 2048, 512 of 4096 NBFM channels (contiguous in frequency)", "c4share"),
        (workloads.CONFIG3, workloads.shard_plan(workloads.config3_plan(), workloads.CONFIG3, 4, 1)[0], 32,
         "config 3, one GPU's share of 4: 20 MS/s cf32, N = 2^19, D = 512, 256 of 1024 channels (NBFM / envelope AM / USB with hang AGC)", "c3share"),
        (workloads.CONFIG5, workloads.config5_plan(), 32,
         "config 5: 256 independent pipelines at 192 kS/s (N = 8192, D = 4), Doppler NCO -> LO2 NCO -> filter -> linear demodulator with PLL + AGC", "c5"),
        (workloads.CONFIG4, workloads.config4_plan(), 16,
         "config 4 with all 4096 NBFM channels on ONE GPU: 64 MS/s cf32, N = 2^20, D = 2048 (north_star's channel count, an eighth of its GPUs)", "c4all"),
    ]
    for shape, plan, nb, label, key in jobs:
        try:
            out.append(measure_shape(pyka9q, torch, device, shape, plan, nb, label, steps=10 if shape.P > 1 else 100, valu_key=key))
        except Exception as e:  # an extra leg must never cost the bench line
            out.append({"workload": label, "error": repr(e)})
    return out


def paced_run(bank, shape, raw, nsrc, blocks, nchan):
    """one block per submit, released at k * L / fs; the PCM of every channel is read back before the block counts as done"""
    pcm = np.empty((nchan, shape.olen), np.int16)
    period = shape.L / shape.fs
    lat, late = [], 0
    t0 = time.perf_counter() + 0.01
    for k in range(blocks):
        due = t0 + k * period
        while True:
            now = time.perf_counter()
            if now >= due:
                break
            time.sleep(min(due - now, 0.002))
        s = bank.stream_slot(0, np.int16, 2 * shape.L)
        s[:] = raw[k % nsrc]
        bank.stream_submit(1)
        bank.pcm_all(1, pcm)
        done = time.perf_counter()
        lat.append(done - due)
        if done > due + period:
            late += 1
    lat = np.array(lat)
    return {"blocks": blocks, "blocks_per_submit": 1, "late_blocks": int(late), "holds_realtime": bool(late == 0),
            "latency_ms_release_to_pcm": {"mean": round(1e3 * float(lat.mean()), 3), "p99": round(1e3 * float(np.percentile(lat, 99)), 3),
                                          "max": round(1e3 * float(lat.max()), 3)},
            "busy_fraction": round(float(lat.mean()) / period, 4),
            "H2D_GBps_while_busy": round(4 * shape.L / float(lat.mean()) / 1e9, 2)}


def realtime_leg(pyka9q, torch, device, shape=None, plan=None, label=None, blocks=1000, full=True):
    """BASELINE metric (i): every channel of a configuration's band, block by block in real time.  The producer writes
    s16le I/Q (IQ_PT, radio.c:159-167) into the library's pinned ring, kh_bank_stream_submit moves it with hipMemcpyAsync
    on the copy stream and runs the launch behind it; the PCM of all channels is read back after every block.
    `late_blocks` counts blocks whose PCM was not back before the next block was due (a backlog).  full: also the unpaced
    PCIe-inclusive rate with the producer's own copy into the ring."""
    shape = shape or workloads.CONFIG2
    plan = plan if plan is not None else workloads.config2_plan(640)
    Tr = 4
    nsrc = 16 if shape.L <= 131072 else 8  # distinct blocks of input, cycled
    x = make_wideband_torch(shape, plan, nsrc * shape.L, device)
    raw = torch.clamp(torch.round(x * (32767.0 * 4.0)), -32768, 32767).to(torch.int16).cpu().numpy().reshape(nsrc, 2 * shape.L)
    del x
    bank = build_bank(pyka9q, shape, plan, Tr, payload=pyka9q.KH_PT_IQ16LE)
    bank.stream_open(4)
    kinds = {}
    for ch in plan:
        kinds[ch["kind"]] = kinds.get(ch["kind"], 0) + 1
    res = {"workload": label or "configs[1] stream, s16le wire format (IQ_PT), all 640 raster channels, pinned host ring -> hipMemcpyAsync "
                                "on a copy stream -> launch; PCM of every channel read back",
           "channels": len(plan), "kinds": kinds, "MSps_realtime": shape.fs / 1e6, "block_ms": round(1e3 * shape.L / shape.fs, 3),
           "hand_over": "host waits for the transfer (KA9Q_HIP_STREAM_HOSTWAIT=1)" if os.environ.get("KA9Q_HIP_STREAM_HOSTWAIT", "1") != "0"
                        else "front stream waits on the transfer's event"}
    # (a) unpaced, Tr blocks per submit: the PCIe-inclusive rate.  The producer's write into the pinned slot is included.
    def unpaced(fill, n):
        nsub = n // Tr
        for j in range(8):
            s = bank.stream_slot(0, np.int16, Tr * 2 * shape.L)
            bank.stream_submit(Tr)
        bank.sync()
        t0 = time.perf_counter()
        for j in range(nsub):
            s = bank.stream_slot(0, np.int16, Tr * 2 * shape.L)
            if fill:
                for b in range(Tr):
                    s[b * 2 * shape.L:(b + 1) * 2 * shape.L] = raw[(j * Tr + b) % nsrc]
            bank.stream_submit(Tr)
        bank.sync()
        return (time.perf_counter() - t0) / (nsub * Tr)
    s = None
    for k in range(4):  # the four ring slots hold distinct data before the "ring pre-filled" pass
        s = bank.stream_slot(0, np.int16, Tr * 2 * shape.L)
        for b in range(Tr):
            s[b * 2 * shape.L:(b + 1) * 2 * shape.L] = raw[(k * Tr + b) % nsrc]
        bank.stream_submit(Tr)
    bank.sync()
    tb = unpaced(False, blocks if full else 200)
    res["pcie_inclusive"] = {"blocks_per_submit": Tr, "us_per_block": round(tb * 1e6, 2), "MSps": round(shape.L / tb / 1e6, 1),
                             "H2D_GBps": round(4 * shape.L / tb / 1e9, 2), "realtime_factor": round((shape.L / shape.fs) / tb, 1),
                             "note": "every submit crosses PCIe (pinned ring -> device raw ring); the producer's own write into the ring is not included"}
    if full:
        tf = unpaced(True, blocks)
        res["pcie_inclusive_with_producer_copy"] = {"us_per_block": round(tf * 1e6, 2), "MSps": round(shape.L / tf / 1e6, 1),
                                                    "note": "as above plus a single-threaded numpy copy of every block into the pinned slot"}
    res["extrapolated_channels_at_realtime"] = int(len(plan) * (shape.L / shape.fs) / tb)
    # (b) paced: one block per submit, released at k * L / fs
    res["paced"] = paced_run(bank, shape, raw, nsrc, blocks, len(plan))
    res["channels_sustained_in_real_time"] = len(plan) if res["paced"]["late_blocks"] == 0 else None
    bank.stream_close()
    bank.close()
    return res


def cpu_baseline(x_host, rots, budget_s=10.0):
    """the oracle (kind 'port': the CPU restatement, the reference's COPTS) on the headline workload, bounded sample:
    one thread, and the box's cores with the channels partitioned over threads behind one shared forward FFT per block
    (SURVEY.md 8(d) CPU baseline plan).  TEST INFRASTRUCTURE used as a reported baseline only."""
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

    try:
        cores = min(16, len(os.sched_getaffinity(0)))  # the box's CPU share for one GPU
    except AttributeError:
        cores = min(16, os.cpu_count() or 1)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    def run_replicas(n, budget):
        """the reference's own way to more cores: several `radio` processes share one multicast I/Q stream, each with its
        own forward FFT and its own channels (README.md:383-390).  Here: n pipelines, the channels dealt over
        them, every one consuming the whole stream on a thread of its own (the C calls release the GIL); the stream
        is sustained at the rate of the slowest replica."""
        from concurrent.futures import ThreadPoolExecutor
        groups = [g for g in np.array_split(np.arange(len(rots)), n) if len(g)]
        pipes = []
        for g in groups:
            pipe = pipeline.Pipeline(L, M, D, FS, compute_n0=True, n0_low=-5500.0, n0_high=5500.0, fast=True, threads=1)
            for j in g:
                pipe.add_channel(demod="fm", rotate=rots[j], low=-5500.0, high=5500.0, kaiser_beta=3.0)
            pipes.append(pipe)
        nb_avail = x_host.size // L
        t0 = time.perf_counter()

        def work(pipe):
            nb = 0
            while time.perf_counter() - t0 < budget:
                k = nb % nb_avail
                pipe.run_block([x_host[k * L:(k + 1) * L]])
                nb += 1
            return nb, time.perf_counter() - t0
        with ThreadPoolExecutor(max_workers=len(pipes)) as pool:
            res = list(pool.map(work, pipes))
        rate = min(nb / dt for nb, dt in res)  # blocks per second every replica has kept up with
        return rate, len(pipes), sum(nb for nb, _ in res), max(dt for _, dt in res)

    nb1, dt1 = run(1, budget_s)
    rate_n, nrep, nbn, dtn = run_replicas(cores, budget_s)
    return {"value": round(nb1 * L / dt1 / 1e6, 3), "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"{nb1} blocks of {L} samples, {C} NBFM channels, {dt1:.1f} s of CPU time",
            "ms_per_block": round(1e3 * dt1 / nb1, 3),
            "all_cores": {"value": round(rate_n * L / 1e6, 3), "unit": "MS/s", "cores": nrep,
                          "sample": f"{nrep} replicas of the pipeline (the reference's model: one `radio` process per group of channels, each "
                                    f"with its own forward FFT), {C // nrep}-{-(-C // nrep)} channels each, all consuming the same stream; "
                                    f"{nbn} blocks in all, {dtn:.1f} s",
                          "ms_per_block": round(1e3 / rate_n, 3)},
            "cpu_model": model, "nproc_machine": os.cpu_count(),
            "compiler": "gcc -O3 -DNDEBUG=1 -funsafe-math-optimizations (the reference's COPTS, Makefile:2, without -march=native); own radix-2/3/5/7 FFT, no FFTW"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the shapes and realtime legs")
    ap.add_argument("--blocks", type=int, default=T, help="blocks per step (per kh_bank_execute call)")
    ap.add_argument("--keep-spectrum", type=int, default=0, choices=[0, 1],
                    help="1: every launch stores the whole spectrum of every block (kh_bank_read_spectrum); 0: only the rows the channels read")
    ap.add_argument("--demod", default="fm", choices=["fm", "none"], help="diagnostic: 'none' runs the filter bank without demodulators")
    args = ap.parse_args()
    globals()["T"] = args.blocks
    globals()["KEEP_SPECTRUM"] = args.keep_spectrum

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a one-GPU box (not a measurement): KA9Q_BENCH_REHEARSE=1 puts every rank on device 0 and runs the
    # collectives over gloo on host tensors, so that the N > 1 control flow can be exercised without N GPUs.
    # Two ranks at most: with four processes on one GPU their streams outnumber the hardware queues and a 70 us step
    # takes milliseconds (seen: the 0.25 s prewarm of one rank did not end within minutes).
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

    verbose = os.environ.get("KA9Q_BENCH_VERBOSE", "0") == "1"

    def note(msg):  # progress on stderr (KA9Q_BENCH_VERBOSE=1): where a multi-rank run is, should one ever stall
        if verbose:
            print(f"[bench rank {rank}] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)

    import pyka9q
    assert pyka9q.lib().ka9q_hip_init(local_rank) == 0, pyka9q.last_error()
    note("library up")

    # 64 channels per GPU on the 12.5 kHz raster: the job's 64*world channels, sharded contiguously in frequency
    full_plan = workloads.config2_plan(C * world)
    plan, owned = workloads.shard_plan(full_plan, SH, world, rank)
    assert len(plan) == C
    rots = [ch["rotate"] for ch in plan]
    x = make_wideband_torch(SH, plan, T * L, device)
    torch.cuda.synchronize()

    bank = pyka9q.Bank(L, M, D, 1, C, T, FS, payload=pyka9q.KH_PT_CF32, compute_n0=1, n0_low=-5500.0, n0_high=5500.0)
    bank.keep_audio(0)
    bank.keep_baseband(0)  # FM: filter output -> discriminator inside one kernel; PCM + status are the outputs
    bank.keep_spectrum(KEEP_SPECTRUM)
    for c, ch in enumerate(plan):
        bank.set_channel(c, demod=pyka9q.KH_DEMOD_FM if args.demod == "fm" else pyka9q.KH_DEMOD_NONE,
                         **workloads.channel_kwargs(ch))
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
    note("prewarm done")
    for _ in range(args.warmup):
        step()
    barrier()
    note("warm-up done, timed region starts")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    note(f"timed region done: {dt:.4f} s")
    # a longer run of the same step (>= 0.5 s), so that the line carries a low-noise figure whatever --steps was
    LONG_S = float(os.environ.get("KA9Q_BENCH_LONG_S", "0.5"))
    n_long, t_long = 0, 0.0
    if LONG_S > 0:
        barrier()
        tl = time.perf_counter()
        while True:
            for _ in range(200):
                step()
            n_long += 200
            bank.sync()
            t_long = time.perf_counter() - tl
            if t_long >= LONG_S:
                break
        if dist is not None:
            tt = torch.tensor([t_long / n_long], device=cdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_long = float(tt.item()) * n_long
    # per-kernel durations, live, with HIP event pairs around every stage: (1) the same K steps with everything on one
    # stream (each kernel alone on the chip), (2) the two-stream pipeline the timed region ran, the events of each stage on
    # the stream it runs on (what the kernel takes beside the other half's kernels)
    nprof = max(args.steps, 20)
    bank.profile(1)
    for _ in range(nprof):
        step()
    bank.sync()
    st = bank.stage_times()
    bank.profile(2)
    for _ in range(nprof):
        step()
    bank.sync()
    st_pipe = bank.stage_times()
    bank.profile(0)

    # SURVEY.md 8(e) / BASELINE configs[3]: every GPU recomputes the forward FFT of the shared stream (option A, what the
    # timed region does) against an RCCL broadcast over xGMI of the step's raw input (option B: saves host fan-out, not
    # the FFT) or of its spectrum (option C: saves the FFT on the peers).  Timed here, outside the timed region.
    probe = None
    note("stage times taken")
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
            def sliced_ms(spec, per_peer):
                """SURVEY.md 8(e) option C-sliced: the root has the step's spectrum and sends every peer only the contiguous
                bin slice its channels read (N / world bins + an N_dec halo, all T blocks): grouped point-to-point sends,
                one per peer, in flight together (on xGMI: one link each)"""
                recv = torch.empty(per_peer, 2, device=cdev, dtype=torch.float32)

                def once():
                    if rank == 0:
                        ops = [dist.P2POp(dist.isend, spec[(r * per_peer) % (spec.shape[0] - per_peer):][:per_peer], r) for r in range(1, world)]
                    else:
                        ops = [dist.P2POp(dist.irecv, recv, 0)]
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                for _ in range(3):
                    once()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                tb = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    once()
                torch.cuda.synchronize()
                v = torch.tensor([(time.perf_counter() - tb) / reps], device=cdev, dtype=torch.float64)
                dist.all_reduce(v, op=dist.ReduceOp.MAX)
                return round(1e3 * float(v.item()), 4)
            raw_like = (x if rank == 0 else torch.empty_like(x)).to(cdev)
            spec_like = torch.empty(T * N, 2, device=cdev, dtype=torch.float32)
            per_peer = T * (N // world + ND)
            probe = {"broadcast_input_ms": bcast_ms(raw_like), "input_bytes": int(raw_like.numel() * 4),
                     "broadcast_spectrum_ms": bcast_ms(spec_like), "spectrum_bytes": int(spec_like.numel() * 4),
                     "sliced_spectrum_ms": sliced_ms(spec_like, per_peer), "sliced_bytes_per_peer": int(per_peer * 8)}
            del spec_like
        except Exception as e:  # the probe must never cost the bench line
            probe = {"error": repr(e)}

    note(f"exchange probe: {probe}")
    # sanity: the run demodulated something (squelch open on the strongest channel)
    stat = bank.status(C - 1, T)
    pcm = bank.pcm(C - 1, T)
    assert args.demod != "fm" or (stat["squelch_open"].max() > 0 and np.abs(pcm).max() > 100), "bench produced no audio"
    x_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:  # (the CPU baseline belongs to the N = 1 line)
        x_cpu = x[:6 * L].cpu().numpy().view(np.complex64).reshape(-1)
    bank.close()
    del x
    torch.cuda.empty_cache()

    if rank == 0:
        bincount = N - (int((N * 5500.0) / FS) - int((N * -5500.0) / FS))
        # compute_n0's second iteration: prepared in the row pass (bracket mode, the default for four-step transforms); with
        # KA9Q_HIP_N0_DIRECT=0 it re-reads the spectrum inside the channel kernel's launch (kh_bank.cpp enqueue_back)
        n0_rides = (ND == 1024 and 2 * OLEN == ND and os.environ.get("KA9Q_HIP_RIDE_N0", "1") != "0"
                    and os.environ.get("KA9Q_HIP_CHAN_WAVE", "1") != "0" and os.environ.get("KA9Q_HIP_N0_DIRECT", "-1") == "0")
        # (the library stores everything unless compute_n0 runs in bracket mode: kh_bank.cpp, masked_now)
        sparse = (not KEEP_SPECTRUM and os.environ.get("KA9Q_HIP_KEEP_SPECTRUM", "0") == "0"
                  and os.environ.get("KA9Q_HIP_N0_DIRECT", "-1") in ("-1", "2"))
        kept = kept_spectrum_bins(N, ND, rots) if sparse else N
        sb = stage_bytes(bincount, n0_rides, kept)
        stages = {}
        # stages whose work rides in another kernel still record an (empty) event pair of ~5 us: no rate for those
        fused_ingest = N > 8192 and os.environ.get("KA9Q_HIP_FUSED_INGEST", "1") != "0"
        empty = {"ingest": "in fft1 (kh_fft_cols reads the raw ring)"} if fused_ingest else {}
        if n0_rides:
            empty["n0"] = "iteration 1 in chan (kh_chan2's launch), final sums in demod (kh_fm_demod2's launch)"
        elif ND == 1024 and 2 * OLEN == ND and os.environ.get("KA9Q_HIP_N0_DIRECT", "-1") != "0":
            empty["n0"] = ("first iteration and the bracket of the second in fft2 (kh_fft_rows), the exact threshold " +
                           "in chan (kh_chan2's launch)" + (" after the block's last row tile has checked the bracket" if sparse else "") +
                           ", final sums in demod (kh_fm_demod2's launch)" if os.environ.get("KA9Q_HIP_N0_DIRECT", "-1") != "1" else
                           "both iterations in fft2 (kh_fft_rows, rendezvous), final sums in demod (kh_fm_demod2's launch)")
        for name, (ms, n) in st.items():
            if n:
                avg = ms / n
                if name in empty:
                    stages[name] = {"ms": round(avg, 5), "GBps": None, "empty": empty[name]}
                else:
                    stages[name] = {"ms": round(avg, 5), "GBps": round(sb[name] / (avg * 1e-3) / 1e9, 1) if avg > 0 else None}
        for name, (ms, n) in st_pipe.items():  # the same stages inside the two-stream pipeline
            if n and name in stages:
                stages[name]["ms_pipelined"] = round(ms / n, 5)
                if name not in empty and ms > 0:
                    stages[name]["GBps_pipelined"] = round(sb[name] / (ms / n * 1e-3) / 1e9, 1)
        # The dominant kernel: the timed run is a two-stream pipeline (kh_bank.cpp: front = ingest, fft1, fft2, n0 of step
        # n + 1 beside back = chan, demod of step n), so a step takes as long as the busier stream; of that stream, the
        # kernel that takes longest there.  (The other stream's kernels run in that one's shadow.)
        piped = lambda k: stages[k].get("ms_pipelined", stages[k]["ms"])
        streams = {"front": [k for k in ("ingest", "fft1", "fft2", "n0") if k in stages],
                   "back": [k for k in ("chan", "demod") if k in stages]}
        stream_ms = {s_: round(sum(piped(k) for k in ks), 5) for s_, ks in streams.items()}
        critical = max(stream_ms, key=stream_ms.get)
        dom = max((k for k in streams[critical] if k not in empty), key=piped)
        traffic = traffic_src = None  # HBM bytes per launch of that kernel from the committed PMC passes (profiles/, tools/summarize_prof.py)
        try:
            # (r0N_stepM_hbm_traffic.json: this command; the other r0N_* files are other shapes or switches)
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_step[0-9]_hbm_traffic.json"))) or \
                sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_step8_hbm_traffic.json")))
            prof = json.load(open(cands[-1]))["kernels"]
            keys = {"fft1": ("kh_fft_cols",), "fft2": ("kh_fft_rows",), "chan": ("kh_chan2", "kh_back_a"),
                    "demod": ("kh_fm_demod2", "kh_fm_finish_w"), "n0": ("kh_n0_pass1",)}.get(dom, ())
            hit = [v for k, v in prof.items() if any(key in k for key in keys)]
            if hit and world == 1 and T == 32:
                traffic = hit[0]["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(cands[-1], ROOT) + " (separate rocprofv3 --pmc passes of this command)"
        except (OSError, ValueError, KeyError, IndexError):
            pass
        dom_ms_alone = stages[dom]["ms"]
        dom_ms = stages[dom].get("ms_pipelined", dom_ms_alone)
        achieved = sb[dom] / (dom_ms * 1e-3) / 1e9
        achieved_alone = sb[dom] / (dom_ms_alone * 1e-3) / 1e9
        kernel_name = pyka9q.lib().kh_stage_kernel_name(pyka9q.KH_STAGES.index(dom)).decode()
        total_alg = SH.algorithmic_bytes(plan, T)  # BASELINE.md section 3 formula
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
                       "blocks_per_step": T, "channels_per_gpu": C,
                       "spectrum": (f"intermediate, not an output: the row pass stores the {kept} of {N} bins per block that lie in rows a channel's "
                                    "filter output reads (kh_bank_keep_spectrum 0); PCM, status and n0 as with the whole spectrum kept "
                                    "(tests/test_gpu_keep_spectrum.py)") if sparse else "all N bins of every block stored",
                       "sharding": "ka9q-radio_amd/shard.py: channels disjoint and contiguous in frequency per GPU, FFT recomputed per GPU",
                       "rank0_channels_k": [int(plan[0]["k"]), int(plan[-1]["k"])]},
            "channel_MS_per_s": round(world * C * args.steps * T * L / dt / 1e6, 1),
            "realtime_factor": round((T * L / FS) / (dt / args.steps), 1),
            "pipeline_GBps_algorithmic": round(total_alg / (ms_per_step * 1e-3) / 1e9, 1),
            "pipeline_frac_of_hbm_peak": round(total_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "stages": stages,
            "value_long": round(world * n_long * T * L / t_long / 1e6, 2) if n_long else None,
            "ms_per_step_long": round(1e3 * t_long / n_long, 4) if n_long else None,
            "steps_long": n_long,
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": sb[dom], "avg_launch_ms": dom_ms,
                         "measured": "HIP events around the kernel on its own stream inside the two-stream pipeline the value was timed in "
                                     "(the longest stage of the stream that bounds a step); *_alone: the same kernel with nothing beside it (one stream)",
                         "critical_stream": critical, "stream_ms_pipelined": stream_ms,
                         "frac_pipelined": round(achieved / HBM_PEAK_GBPS, 4), "frac_alone": round(achieved_alone / HBM_PEAK_GBPS, 4),
                         "achieved_alone": round(achieved_alone, 1), "avg_launch_ms_alone": dom_ms_alone},
        }
        if probe is not None:
            fft_ms = sum(stages[k]["ms"] for k in ("ingest", "fft1", "fft2") if k in stages)
            probe["recompute_forward_fft_ms"] = round(fft_ms, 4)
            if "broadcast_spectrum_ms" in probe:
                # north_star: "whichever rocprof and RCCL timing show to be cheaper".  Options that replace the per-GPU
                # transform: the full spectrum broadcast or the per-peer slices (option C); the input broadcast (option B)
                # only replaces the host's fan-out and still needs the transform on every GPU.
                cands = {"recompute per GPU": fft_ms, "spectrum broadcast": probe["broadcast_spectrum_ms"],
                         "sliced spectrum (grouped send/recv)": probe.get("sliced_spectrum_ms", float("inf")),
                         "input broadcast + recompute": probe["broadcast_input_ms"] + fft_ms}
                probe["options_ms"] = {k: round(v, 4) for k, v in cands.items()}
                probe["cheaper"] = min(cands, key=cands.get)
            out["exchange_probe"] = probe
        if world == 1 and not args.no_extra:
            out["shapes"] = shapes_leg(pyka9q, torch, device)
            try:
                out["realtime"] = realtime_leg(pyka9q, torch, device)
            except Exception as e:
                out["realtime"] = {"error": repr(e)}
            # north_star's target shape on the hardware there is: all 4096 NBFM channels of the 64 MS/s stream on ONE GPU,
            # paced at the stream's own block rate (radio.c:132-204, sizing main.c:326-330); and config 3's 1024 mixed channels
            for key, shape, plan, label in (
                    ("realtime_config4", workloads.CONFIG4, workloads.config4_plan(),
                     "config 4 stream: 64 MS/s s16le (IQ_PT), N = 2^20, all 4096 NBFM 12.5 kHz channels on one GPU, pinned host ring -> "
                     "hipMemcpyAsync -> launch, one 8.192 ms block per submit; PCM of every channel read back"),
                    ("realtime_config3", workloads.CONFIG3, workloads.config3_plan(),
                     "config 3 stream: 20 MS/s s16le, N = 2^19, all 1024 mixed channels (NBFM / envelope AM / USB) on one GPU, one "
                     "13.1 ms block per submit; PCM of every channel read back")):
                try:
                    out[key] = realtime_leg(pyka9q, torch, device, shape, plan, label, blocks=1000, full=False)
                except Exception as e:
                    out[key] = {"error": repr(e)}
        if x_cpu is not None:
            out["cpu_baseline"] = cpu_baseline(x_cpu, rots)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
