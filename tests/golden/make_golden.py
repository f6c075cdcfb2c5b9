#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz from the REAL reference objects in oracle/_ref (osc.c, decimate.c, misc.c
and ax25.c
compiled unmodified from /root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

The fixtures hold inputs and the reference's outputs, nothing else.  tests/test_oracle_golden.py holds the CPU
oracle to them wherever the tests run (the reference itself does not travel to the GPU box).
filter.c / fm.c / linear.c / am.c / audio.c / radio.c do not build here (<fftw3.h>, <bsd/string.h> absent), so
there are no fixtures for them: their parity is unpinned (DESIGN.md "Oracle")."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402


def osc_cases():
    # (freq cycles/sample, rate cycles/sample^2, steps): LO2 at -48 kHz of 192 kHz, Doppler sweeps (radio.c:233-240),
    # a retune in mid-run (phase kept, osc.c:24-27), long runs across Renorm_rate = 16384 (osc.c:11)
    return [(-0.25, 0.0, 40000), (0.123456789, 0.0, 40000), (10e3 / 192e3, -(200.0 / 192e3) / 192e3, 70000),
            (-7e3 / 192e3, (150.0 / 192e3) / 192e3, 70000), (1e-7, 0.0, 20000), (0.0, 0.0, 100)]


def gen_osc(R):
    out = {}
    for i, (f, r, n) in enumerate(osc_cases()):
        o = orc.RefOsc()
        R.set_osc(C.addressof(o), f, r)
        ph = np.empty((n, 2), np.float64)
        for k in range(n):
            v = R.step_osc(C.addressof(o))
            ph[k] = (v.re, v.im)
        # retune with phase continuity, then 3000 more steps
        R.set_osc(C.addressof(o), -f * 0.5 + 0.01, 0.0)
        ph2 = np.empty((3000, 2), np.float64)
        for k in range(3000):
            v = R.step_osc(C.addressof(o))
            ph2[k] = (v.re, v.im)
        idx = np.unique(np.concatenate([np.arange(0, 64), np.arange(16380, min(16390, n)), np.arange(n - 64, n),
                                        np.arange(0, n, 997)]))
        idx = idx[idx < n]
        out[f"case{i}_params"] = np.array([f, r, n], np.float64)
        out[f"case{i}_idx"] = idx.astype(np.int64)
        out[f"case{i}_phasor"] = ph[idx]
        out[f"case{i}_retuned"] = ph2[::100]
        out[f"case{i}_steps_after"] = np.array([o.steps], np.int64)
    return out


def gen_pll(R):
    out = {}
    rng = np.random.default_rng(20181)
    for i, (bw, damping, fs, n) in enumerate([(5.0, np.sqrt(0.5), 48000.0, 6000), (50.0, 1.0, 31250.0, 6000)]):
        p = orc.RefPll()
        R.init_pll(C.addressof(p), bw, damping, 0.0, fs)
        # a phase detector output that looks like a 3 Hz offset being pulled in, plus noise
        phase = (0.3 * np.sin(2 * np.pi * 3.0 * np.arange(n) / fs) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        fb = np.empty(n, np.float32)
        vco = np.empty((n, 2), np.float64)
        for k in range(n):
            fb[k] = R.run_pll(C.addressof(p), float(phase[k]))
            vco[k] = (p.vco.phasor.re, p.vco.phasor.im)
        out[f"pll{i}_params"] = np.array([bw, damping, fs], np.float64)
        out[f"pll{i}_gains"] = np.array([p.integrator_gain, p.prop_gain, p.samptime], np.float32)
        out[f"pll{i}_phase"] = phase
        out[f"pll{i}_feedback"] = fb
        out[f"pll{i}_vco"] = vco[::50]
        out[f"pll{i}_integrator"] = np.array([p.integrator], np.float32)
    return out


def f8_state():
    s = orc.Hb15()
    # Goodman/Carey F8 as hackrf.c:442-451 scales it
    for j, c in enumerate((-6.0, 33.0, -116.0, 490.0)):
        s.coeffs[j] = c / 802.0
    return s


def gen_decimate(R, tag):
    out = {}
    rng = np.random.default_rng(0x6B61)
    chunks = [1000, 8, 3000, 24, 1]  # hb15's SSE path needs nothing special; odd and tiny counts carry state too
    x = rng.standard_normal(2 * sum(chunks)).astype(np.float32)
    s = f8_state()
    hb3_state = C.c_float(0.0)
    y15, y3 = [], []
    pos = 0
    for cnt in chunks:
        seg = x[pos:pos + 2 * cnt].copy()
        o = np.empty(cnt, np.float32)
        R.hb15_block(C.addressof(s), o.ctypes.data_as(C.c_void_p), seg.ctypes.data_as(C.c_void_p), cnt)
        y15.append(o)
        seg = x[pos:pos + 2 * cnt].copy()
        o = np.empty(cnt, np.float32)
        R.hb3_block(C.addressof(hb3_state), o.ctypes.data_as(C.c_void_p), seg.ctypes.data_as(C.c_void_p), cnt)
        y3.append(o)
        pos += 2 * cnt
    out["chunks"] = np.array(chunks, np.int64)
    out["x"] = x
    out[f"hb15_{tag}"] = np.concatenate(y15)
    out[f"hb3_{tag}"] = np.concatenate(y3)
    out[f"hb15_state_{tag}"] = np.array(list(s.even_samples) + list(s.odd_samples) + list(s.old_odd_samples), np.float32)
    out[f"hb3_state_{tag}"] = np.array([hb3_state.value], np.float32)
    # in place (output == input), as hackrf.c:497-509 calls it
    seg = x[:2 * 512].copy()
    s2 = f8_state()
    R.hb15_block(C.addressof(s2), seg.ctypes.data_as(C.c_void_p), seg.ctypes.data_as(C.c_void_p), 512)
    out[f"hb15_inplace_{tag}"] = seg[:512].copy()
    return out


def gen_misc(R):
    R.nextfastfft.restype = C.c_uint32
    R.nextfastfft.argtypes = [C.c_uint32]
    ns = [1, 2, 7, 100, 7679, 8191, 8192, 319999, 320000, 799999, 1048575, 2 ** 20, 123456789, 4288306049, 4288306050]
    return {"n": np.array(ns, np.uint64), "next": np.array([R.nextfastfft(n) for n in ns], np.uint64)}


def gen_ax25(R):
    """crc_good (ax25.c:140-156) on frames with a correct CRC (built by synth.ax25_crc, an independent
    implementation), frames with one flipped bit, and random bytes."""
    sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
    import synth
    R.crc_good.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(0xA25)
    rows, lens, good = [], [], []
    for k in range(60):
        n = int(rng.integers(1, 300))
        body = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        f = bytearray(body + synth.ax25_crc(body))
        if k % 3 == 1:
            f[int(rng.integers(0, len(f)))] ^= 1 << int(rng.integers(0, 8))
        elif k % 3 == 2:
            f = bytearray(rng.integers(0, 256, n + 2, dtype=np.uint8).tobytes())
        b = np.frombuffer(bytes(f), np.uint8).copy()
        row = np.zeros(302, np.uint8)
        row[:b.size] = b
        rows.append(row)
        lens.append(b.size)
        good.append(int(R.crc_good(b.ctypes.data_as(C.c_void_p), b.size)))
    return {"data": np.stack(rows), "len": np.array(lens, np.int64), "good": np.array(good, np.int64)}


def main():
    R = orc.ref()
    Rp = orc.ref(portable=True)
    if R is None or Rp is None:
        sys.exit("oracle/_ref is missing: run `make -C oracle` where /root/reference exists")
    np.savez_compressed(os.path.join(HERE, "ref_osc.npz"), **gen_osc(R), **gen_pll(R))
    d = gen_decimate(R, "sse")
    d.update(gen_decimate(Rp, "portable"))
    np.savez_compressed(os.path.join(HERE, "ref_decimate.npz"), **d)
    np.savez_compressed(os.path.join(HERE, "ref_misc.npz"), **gen_misc(R))
    np.savez_compressed(os.path.join(HERE, "ref_ax25.npz"), **gen_ax25(R))
    for f in ("ref_osc.npz", "ref_decimate.npz", "ref_misc.npz", "ref_ax25.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
