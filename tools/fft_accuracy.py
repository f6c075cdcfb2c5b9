#!/usr/bin/env python3
"""forward transform accuracy against numpy's complex128 FFT: relative L2 error of the whole spectrum, and the error
floor in bins far from a strong carrier relative to the carrier's bin (what a quiet channel next to strong ones sees).
   python3 tools/fft_accuracy.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import pyka9q as hip, orc
assert hip.lib().ka9q_hip_init(0) == 0
rng = np.random.default_rng(3)
for logn in (13, 16, 18, 19, 20):
    N = 1 << logn
    L = N // 2
    M = N - L + 1
    t = np.arange(N)
    x = (0.2 * np.exp(2j * np.pi * (N // 8 + 0.3) / N * t) + 1e-3 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    X = np.fft.fft(x.astype(np.complex128))
    b = hip.Bank(L, M, max(N // 1024, 4), 1, 1, 2, 8_000_000)
    b.set_channel(0, demod=hip.KH_DEMOD_NONE, rotate=0, low=-3000.0, high=3000.0, kaiser_beta=3.0)
    b.write_input(0, x[M - 1:])          # block 0: history zero + first L samples -> window = [0.., x[M-1:]]: not x
    b.execute(1)
    # feed so that the second block's window is exactly x: first launch supplies x[:L]... simpler: compare block with its own window
    w = np.zeros(N, np.complex64)
    w[M - 1:] = x[M - 1:]
    Xw = np.fft.fft(w.astype(np.complex128))
    G = b.spectrum(0, 0).astype(np.complex128)
    b.close()
    fin = orc.lib().orc_create_filter_input(L, M, orc.COMPLEX)
    err = np.abs(G - Xw)
    far = np.ones(N, bool)
    k0 = N // 8
    far[max(0, k0 - N // 16):k0 + N // 16] = False
    print(f"N=2^{logn}: rel L2 error {np.linalg.norm(G - Xw) / np.linalg.norm(Xw):.2e}; error floor far from the carrier "
          f"{np.sqrt((err[far] ** 2).mean()) / np.abs(Xw).max():.2e} of the carrier's bin (fp32 eps 6e-8)", flush=True)
