"""Deterministic synthetic I/Q for the tests and bench.py: counter-based Gaussian noise
plus per-channel NBFM / AM / SSB signals summed at the wideband rate.  numpy only (the bench regenerates the
same shapes on the GPU with torch when it needs tens of millions of samples)."""
import sys

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


# ---------------------------------------------------------------- AX.25 / AFSK-1200 test signals (for packet.c's decoder)
def ax25_crc(data):
    """CRC-CCITT as AX.25 sends it (reflected 0x8408, preset 0xffff, complemented, low byte first)."""
    crc = 0xFFFF
    for byte in data:
        for i in range(8):
            fb = 0x8408 if ((crc ^ (byte >> i)) & 1) else 0
            crc = (crc >> 1) ^ fb
    crc ^= 0xFFFF
    return bytes([crc & 0xFF, crc >> 8])


def ax25_ui_frame(dest, src, info, dest_ssid=0, src_ssid=0):
    """A UI frame (control 0x03, PID 0xf0) with its CRC appended: the bytes between the flags."""
    def addr(call, ssid, last):
        call = (call.upper() + "      ")[:6]
        return bytes([ord(c) << 1 for c in call]) + bytes([0x60 | ((ssid & 15) << 1) | (1 if last else 0)])
    body = addr(dest, dest_ssid, False) + addr(src, src_ssid, True) + bytes([0x03, 0xF0]) + bytes(info)
    return body + ax25_crc(body)


def hdlc_bits(frames, preamble_flags=8, between_flags=2, tail_flags=2):
    """Flags + bit-stuffed frames, least significant bit first; returns the NRZ data bits (1 = no transition)."""
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    bits = flag * preamble_flags
    for k, f in enumerate(frames):
        ones = 0
        for byte in f:
            for i in range(8):
                b = (byte >> i) & 1
                bits.append(b)
                if b:
                    ones += 1
                    if ones == 5:
                        bits.append(0)
                        ones = 0
                else:
                    ones = 0
        bits += flag * (between_flags if k + 1 < len(frames) else tail_flags)
    return bits


def afsk1200(bits, fs=48000.0, baud=1200.0, amp=0.5, lead=0, tail=0, mark=1200.0, space=2200.0, phase0=0.0):
    """NRZI (a 0 toggles the tone, a 1 keeps it), continuous-phase FSK; `lead`/`tail` samples of silence around it."""
    spb = fs / baud
    n = int(round(len(bits) * spb))
    tone = np.empty(n)
    cur = mark
    edges = [int(round(k * spb)) for k in range(len(bits) + 1)]
    for k, b in enumerate(bits):
        if not b:
            cur = space if cur == mark else mark
        tone[edges[k]:edges[k + 1]] = cur
    ph = phase0 + 2 * np.pi * np.cumsum(tone) / fs
    x = amp * np.sin(ph)
    return np.concatenate([np.zeros(lead), x, np.zeros(tail)])


def pcm_be(x):
    """float -> big-endian int16 words the way audio.c:22-28,44 sends them (truncation toward zero, saturation)."""
    v = np.clip(np.trunc(32767.0 * np.asarray(x, np.float64)), -32768, 32767).astype(np.int16)
    return v.byteswap() if sys.byteorder == "little" else v
