"""Deterministic synthetic I/Q for the tests and bench.py: counter-based Gaussian noise
plus per-channel NBFM / AM / SSB signals summed at the wideband rate.  numpy only (the bench regenerates the
same shapes on the GPU with torch when it needs tens of millions of samples)."""
import numpy as np

SEED = 0x6B613971


def noise(nsamples, sigma=0.01, stream=0, seed=SEED):
    """complex white Gaussian noise, sigma per rail, from Philox keyed by (seed, stream)."""
    rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, stream]))
    x = rng.standard_normal((nsamples, 2), dtype=np.float32) * np.float32(sigma)
    return x.view(np.complex64).reshape(nsamples)


def rotate_for(k, spacing_hz, N, fs):
    """bin index of channel k on a `spacing_hz` raster, rounded to an even bin: with
    L = N/2 an even rotate keeps the block-to-block phase continuous (filter.c:270-289 never corrects it)."""
    return int(2 * round(k * spacing_hz * N / fs / 2))


def fm_carrier(n0, nsamples, fs, f_carrier, amp, f_tone, deviation, phase0=0.0):
    """NBFM carrier at f_carrier: amp * exp(j(2 pi f_c t + (dev/f_tone) sin(2 pi f_tone t)))."""
    t = (np.arange(n0, n0 + nsamples, dtype=np.float64)) / fs
    ph = 2 * np.pi * f_carrier * t + (deviation / f_tone) * np.sin(2 * np.pi * f_tone * t) + phase0
    return (amp * np.exp(1j * ph)).astype(np.complex64)


def am_carrier(n0, nsamples, fs, f_carrier, amp, f_tone, depth, phase0=0.0):
    t = (np.arange(n0, n0 + nsamples, dtype=np.float64)) / fs
    env = amp * (1.0 + depth * np.cos(2 * np.pi * f_tone * t))
    return (env * np.exp(1j * (2 * np.pi * f_carrier * t + phase0))).astype(np.complex64)


def usb_two_tone(n0, nsamples, fs, f_carrier, amp, f1=700.0, f2=1900.0):
    t = (np.arange(n0, n0 + nsamples, dtype=np.float64)) / fs
    x = np.exp(2j * np.pi * (f_carrier + f1) * t) + np.exp(2j * np.pi * (f_carrier + f2) * t)
    return (0.5 * amp * x).astype(np.complex64)


def swept_carrier(n0, nsamples, fs, f0, fdot, amp, f_tone=400.0, depth=0.5, phase0=0.0):
    """AM (DSB with carrier) signal whose carrier starts at f0 Hz (at sample 0) and sweeps fdot Hz/s."""
    t = (np.arange(n0, n0 + nsamples, dtype=np.float64)) / fs
    ph = 2 * np.pi * (f0 * t + 0.5 * fdot * t * t) + phase0
    env = amp * (1.0 + depth * np.cos(2 * np.pi * f_tone * t))
    return (env * np.exp(1j * ph)).astype(np.complex64)


def wideband_fm(n0, nsamples, fs, N, channels, spacing_hz=12500.0, sigma=0.01, stream=0):
    """C2-style wideband block: noise + one NBFM carrier per (k, amp, tone, deviation) at the centre
    frequency of bin rotate_for(k, ...)."""
    x = noise(nsamples, sigma, stream)
    for (k, amp, tone, dev) in channels:
        fc = rotate_for(k, spacing_hz, N, fs) * fs / N
        x = x + fm_carrier(n0, nsamples, fs, fc, amp, tone, dev, phase0=0.1 * k)
    return x.astype(np.complex64)


def to_int16(x, scale=32767.0):
    """quantise cf32 to interleaved int16 I/Q (round to nearest), as a front end would."""
    v = np.empty(2 * x.size, np.float32)
    v[0::2] = x.real
    v[1::2] = x.imag
    return np.clip(np.rint(v * scale), -32768, 32767).astype(np.int16)
