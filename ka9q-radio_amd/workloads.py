"""Channel plans of the BASELINE.json configurations ("Synthetic inputs"), shared by bench.py, the GPU
tests and the multi-process tests: which channels exist, what each one demodulates, and which rank of a multi-GPU job
serves it.  Plans are plain data (lists of dicts with kh_bank_set_channel's keyword arguments); nothing here touches a
GPU.  The parameter values are the reference's own: the FM/AM/USB presets of modes.txt:26-39, main.c:132-146's defaults
(headroom -15 dB, AGC start gain +80 dB, recovery +6 dB/s), modes.txt:37's hang time of 1.1 s."""
import numpy as np

import shard
import synth

HEADROOM = 10 ** (-15 / 20)   # main.c:136
AGC0 = 10 ** (80 / 20)        # main.c:139
RASTER = 12500.0


class Shape:
    """Sizes of one configuration: main.c:326-330's L/M/N with the power-of-two variants of SURVEY.md 8."""

    def __init__(self, name, fs, L, M, D, n_inputs=1):
        self.name, self.fs, self.L, self.M, self.D, self.P = name, fs, L, M, D, n_inputs
        self.N = L + M - 1
        self.ND = self.N // D
        self.olen = L // D
        self.out_rate = fs / D

    def algorithmic_bytes(self, plan, nblocks):
        """B = 8N (time buffer) + 8N (spectrum) + sum over channels of 8 N_dec (bins) + 8 N_dec (response)
        + b_out * olen, per block and input; b_out = 2 (mono int16) or 4 (stereo)."""
        per_block = self.P * 16 * self.N
        for ch in plan:
            b_out = 4 if ch.get("channels", 1) == 2 else 2
            per_block += 16 * self.ND + b_out * self.olen
        return nblocks * per_block


CONFIG1 = Shape("config 1: 192 kS/s, 1 NBFM channel", 192_000, 3840, 4353, 4)
CONFIG2 = Shape("configs[1]: 8 MS/s wideband, NBFM 12.5 kHz channels", 8_000_000, 131072, 131073, 256)
CONFIG3 = Shape("config 3: 20 MS/s wideband, mixed AM/FM/SSB channels", 20_000_000, 262144, 262145, 512)
CONFIG4 = Shape("config 4: 64 MS/s, 2^20-point blocks, NBFM channels", 64_000_000, 524288, 524289, 2048)
CONFIG5 = Shape("config 5: 256 pipelines at 192 kS/s, linear + PLL + AGC under Doppler", 192_000, 3840, 4353, 4, n_inputs=256)


def _recovery(out_rate):
    return 10 ** ((6.0 / out_rate) / 20)  # +6 dB/s as a per-sample ratio (radio.c:398)


def fm_channel(shape, k, j=0):
    """NBFM as C2: +-5.5 kHz at the output rate, Kaiser beta 3; amplitude 0.01 .. 0.1 log-spaced over j = 0..63"""
    return dict(k=int(k), kind="fm", demod="fm", rotate=synth.rotate_for(k, RASTER, shape.N, shape.fs), low=-5500.0, high=5500.0,
                kaiser_beta=3.0, headroom=HEADROOM, amp=0.01 * (10.0 ** ((j % 64) / 63.0)), tone=300.0 + 50.0 * (j % 64),
                deviation=2500.0)


def am_channel(shape, k, j=0):
    """envelope AM (modes.txt AM preset: `env`, AGC on), +-5 kHz; carrier 0.05, 30-80 % modulation"""
    return dict(k=int(k), kind="am", demod="linear", rotate=synth.rotate_for(k, RASTER, shape.N, shape.fs), low=-5000.0, high=5000.0,
                kaiser_beta=3.0, env=1, agc=1, hangtime=1.1 * shape.out_rate, recovery_rate=_recovery(shape.out_rate),
                agc_gain=AGC0, headroom=HEADROOM, amp=0.05, tone=400.0, depth=0.3 + 0.5 * ((j % 11) / 10.0))


def usb_channel(shape, k, j=0):
    """USB (modes.txt:37): +100 .. +3000 Hz, AGC recovery +6 dB/s, hang 1.1 s; two-tone 700/1900 Hz"""
    return dict(k=int(k), kind="usb", demod="linear", rotate=synth.rotate_for(k, RASTER, shape.N, shape.fs), low=100.0, high=3000.0,
                kaiser_beta=3.0, agc=1, hangtime=1.1 * shape.out_rate, recovery_rate=_recovery(shape.out_rate), agc_gain=AGC0,
                headroom=HEADROOM, amp=0.05)


def raster_fm_plan(shape, ks):
    return [fm_channel(shape, k, j) for j, k in enumerate(ks)]


def config2_plan(n_channels=64):
    """C2: NBFM carriers on the 12.5 kHz raster, k = -n/2 .. n/2-1 (64 in BASELINE.json; 640 fill the 8 MHz)."""
    return raster_fm_plan(CONFIG2, range(-(n_channels // 2), n_channels - n_channels // 2))


def mixed_plan(shape, n_channels):
    """C3's rule on any shape: k mod 4 = 0, 1 -> NBFM; 2 -> envelope AM; 3 -> USB; k = -n/2 .. n/2-1"""
    plan = []
    for j, k in enumerate(range(-(n_channels // 2), n_channels - n_channels // 2)):
        m = k % 4
        plan.append(fm_channel(shape, k, j) if m in (0, 1) else am_channel(shape, k, j) if m == 2 else usb_channel(shape, k, j))
    return plan


def config3_plan(n_channels=1024):
    return mixed_plan(CONFIG3, n_channels)


def plan_signal(plan, shape, n0, nsamples, sigma=0.01):
    """the wideband input a plan describes: noise + one FM / AM / two-tone USB signal per channel at its bin centre"""
    x = synth.noise(nsamples, sigma).astype(np.complex128)
    for ch in plan:
        fc = ch["rotate"] * shape.fs / shape.N
        if ch["kind"] == "fm":
            x += synth.fm_carrier(n0, nsamples, shape.fs, fc, ch["amp"], ch["tone"], ch["deviation"], phase0=0.1 * ch["k"])
        elif ch["kind"] == "am":
            x += synth.am_carrier(n0, nsamples, shape.fs, fc, ch["amp"], ch["tone"], ch["depth"], phase0=0.3)
        elif ch["kind"] == "usb":
            x += synth.usb_two_tone(n0, nsamples, shape.fs, fc, ch["amp"])
    return x.astype(np.complex64)


def config4_plan(n_channels=4096):
    """C4: NBFM on the raster, k = -2048 .. 2047"""
    return raster_fm_plan(CONFIG4, range(-(n_channels // 2), n_channels - n_channels // 2))


def config5_plan(n_pipelines=256, seed=5):
    """C5: every pipeline its own input at 192 kS/s with a DSB-with-carrier (CAM: pll) or suppressed-carrier (pll +
    square) signal at +48 kHz IF whose carrier is swept by f0 in +-10 kHz, fdot in +-200 Hz/s; the receiver applies the
    matching open-loop correction (set_doppler, radio.c:233-240: both negated) and LO2 = -48 kHz."""
    rng = np.random.default_rng(seed)
    plan = []
    sh = CONFIG5
    for p in range(n_pipelines):
        f0, fdot = float(rng.uniform(-10e3, 10e3)), float(rng.uniform(-200.0, 200.0))
        plan.append(dict(k=p, kind="cam" if p % 2 == 0 else "dsb", demod="linear", input=p, rotate=0, low=-5000.0, high=5000.0,
                         kaiser_beta=3.0, pll=1, square=int(p % 2), agc=1, hangtime=1.1 * sh.out_rate,
                         recovery_rate=_recovery(sh.out_rate), agc_gain=AGC0, headroom=HEADROOM, loop_bw=5.0, amp=0.05,
                         if_hz=48000.0, f0=f0, fdot=fdot,
                         lo2=-48000.0 / sh.fs, doppler=-f0 / sh.fs, doppler_rate=-fdot / (sh.fs * sh.fs)))
    return plan


SET_CHANNEL_KEYS = ("input", "rotate", "out_type", "low", "high", "kaiser_beta", "flat", "headroom", "pll", "square", "agc", "env",
                    "channels", "hangtime", "recovery_rate", "agc_gain", "loop_bw", "shift")


def channel_kwargs(ch):
    """what kh_bank_set_channel / pipeline.add_channel take (the signal description keys stay behind)"""
    return {k: ch[k] for k in SET_CHANNEL_KEYS if k in ch}


def shard_plan(plan, shape, world_size, rank):
    """rank `rank`'s channels of a wideband plan -- contiguous in frequency, equal counts to within one
    (shard.partition).  Independent pipelines (config 5: one input per channel) shard by pipeline index."""
    if shape.P > 1:
        groups = np.array_split(np.arange(len(plan)), world_size)
        idx = groups[rank]
    else:
        idx = shard.partition([ch["rotate"] for ch in plan], shape.N, world_size)[rank]
    return [plan[i] for i in idx], [int(i) for i in idx]
