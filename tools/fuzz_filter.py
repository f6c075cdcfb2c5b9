#!/usr/bin/env python3
"""Randomised parity sweep of the filter.h drop-in ABI against the CPU oracle (test infrastructure, run on a GPU box):
tools/fuzz_filter.py [cases] [seed].  Draws (N, M, D, types, rotate, passband) at random among the shapes the library
accepts -- powers of two up to 2^19 and 2,3,5,7-smooth sizes up to 8192 -- fills the overlap-save history, then compares two blocks and reports every
case above the 1e-5 relative-RMS bar.  Exit status 1 if any case fails."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import orc  # noqa: E402
import pyka9q as hip  # noqa: E402


def rel_rms(a, b):
    a = np.asarray(a).astype(np.complex128)
    b = np.asarray(b).astype(np.complex128)
    d = np.sqrt(np.mean(np.abs(a - b) ** 2))
    return d / max(np.sqrt(np.mean(np.abs(b) ** 2)), 1e-30)


def smooth_sizes(limit):
    out = []
    for n in range(16, limit + 1, 2):
        m = n
        for r in (2, 3, 5, 7):
            while m % r == 0:
                m //= r
        if m == 1:
            out.append(n)
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    small = smooth_sizes(8192)
    bad = 0
    t0 = time.time()
    for k in range(cases):
        if rng.random() < 0.5:
            N = int(small[rng.integers(len(small))])
        else:
            N = 1 << int(rng.integers(8, 20))
        divs = [d for d in range(1, 4097) if N % d == 0 and (N // d) % 2 == 0 and N // d >= 16]
        D = int(divs[rng.integers(len(divs))])
        # L must be a multiple of D (filter.c:171-173 truncates otherwise); M = N - L + 1 anywhere in (1, N)
        lo, hi = max(1, N // D // 24), N // D - 1  # (at most ~26 blocks until the history has filled)
        L = D * int(rng.integers(lo, hi + 1))
        M = N - L + 1
        in_type = orc.COMPLEX if N > 8192 or rng.random() < 0.6 else orc.REAL
        out_type = [orc.COMPLEX, orc.CROSS_CONJ, orc.REAL][int(rng.integers(3))]
        rotate = 0 if in_type == orc.REAL else int(rng.integers(-N + 1, N))
        lowf, highf = sorted(rng.uniform(-0.5, 0.5, 2))
        if out_type == orc.REAL or in_type == orc.REAL:
            lowf, highf = abs(lowf) * 0.9 + 0.005, abs(lowf) * 0.9 + 0.1
            highf = min(highf, 0.49)
        tag = f"N={N} L={L} M={M} D={D} in={in_type} out={out_type} rot={rotate} pass=({lowf:.3f},{highf:.3f})"
        try:
            g = hip.Filter(L, M, D, in_type, out_type)
        except Exception as e:  # a shape the library refuses: say so, it is not a parity failure
            print(f"[{k}] refused {tag}: {e}", flush=True)
            continue
        o = orc.Filter(L, M, D, in_type, out_type)
        g.set_filter(lowf, highf, 3.0)
        o.set_filter(lowf, highf, 3.0)
        # The history holds M-1 samples: until real input has filled it the kept L/D outputs are the filter's start-up
        # transient, orders of magnitude below the transform's intermediate values, and their relative error says
        # nothing (fp32 on both sides).  Fill first, then compare two blocks.
        fill = -(-(M - 1) // L)
        olen_ = L // D
        # (a relative RMS needs samples to be one: with one or two outputs per block a single value that happens to lie near
        # zero decides it -- N=1280 L=40 D=40, REAL to REAL: 7e-5 on two samples.  At least 64 outputs are compared, and
        # blocks shorter than that are judged together)
        nb = fill + max(2, -(-64 // olen_))
        x = (rng.standard_normal(nb * L) + 1j * rng.standard_normal(nb * L)).astype(np.complex64) * np.float32(0.1)
        if in_type == orc.REAL:
            x = x.real.copy()
        resp_err = worst = rel_rms(g.response(), o.response())
        ygs, yos = [], []
        for b in range(nb):
            blk = x[b * L:(b + 1) * L]
            yg, yo = g.execute(blk, rotate), o.execute(blk, rotate)
            if b >= fill:
                if olen_ >= 64:
                    worst = max(worst, rel_rms(yg, yo))
                else:
                    ygs.append(np.array(yg))
                    yos.append(np.array(yo))
        if ygs:
            worst = max(worst, rel_rms(np.concatenate(ygs), np.concatenate(yos)))
        ok = worst < 1e-5
        bad += not ok
        print(f"[{k}] {'ok ' if ok else 'BAD'} {worst:.2e} (response {resp_err:.2e}) {tag}", flush=True)
        g.close()
        o.close()
    print(f"{cases} cases, {bad} bad, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
