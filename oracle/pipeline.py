"""The reference's receive chain assembled from the CPU oracle, block by block, in the reference's order:
proc_samples (radio.c:132-204) -> execute_filter_input (filter.c:212) -> per channel execute_filter_output
(filter.c:240) -> demod_fm / demod_linear / demod_am (fm.c, linear.c, am.c) -> scaleclip (audio.c:22).

TEST INFRASTRUCTURE ONLY (checker and bench.py's cpu_baseline leg)."""
import ctypes as C

import numpy as np

import orc


class Channel:
    def __init__(self, pipe, rotate=0, out_type=orc.COMPLEX, demod="none", low=-5000.0, high=5000.0,
                 kaiser_beta=3.0, flat=0, headroom=10 ** (-15 / 20), pll=0, square=0, agc=0, env=0, channels=1,
                 hangtime=0.0, recovery_rate=1.0, agc_gain=1.0, loop_bw=5.0, shift=0.0, input=0, **_):
        self.rotate = rotate
        self.demod_kind = demod
        self.input = input
        out_rate = np.float32(pipe.samprate) / np.float32(pipe.D)
        self.fout = orc.lib(pipe.fast).orc_create_filter_output(pipe.fins[input].fin, None, pipe.D, out_type)
        self.lib = orc.lib(pipe.fast)
        # radio.c:408-411
        assert self.lib.orc_set_filter(self.fout, float(np.float32(low) / out_rate), float(np.float32(high) / out_rate),
                                       kaiser_beta) == 0
        self.olen = pipe.L // pipe.D
        out_samprate = int(out_rate)
        if demod == "fm":
            self.dem = orc.FmDemod(pipe.samprate, pipe.D, headroom, low, high, flat, fast=pipe.fast)
        elif demod == "linear":
            self.dem = orc.LinearDemod(out_samprate, pll, square, agc, env, channels, headroom, hangtime,
                                       recovery_rate, loop_bw, agc_gain, shift, fast=pipe.fast)
        elif demod == "am":
            self.dem = orc.AmDemod(pipe.samprate, pipe.D, out_samprate, headroom, hangtime, recovery_rate, agc_gain,
                                   fast=pipe.fast)
        else:
            self.dem = None

    def _out(self):
        p = self.lib.orc_filter_out_output(self.fout)
        buf = (C.c_char * (self.olen * 8)).from_address(p)
        return np.frombuffer(buf, dtype=np.complex64, count=self.olen)

    def response(self, nd):
        p = self.lib.orc_filter_out_response(self.fout)
        buf = (C.c_char * (nd * 8)).from_address(p)
        return np.frombuffer(buf, dtype=np.complex64, count=nd).copy()

    def noise_gain(self):
        return self.lib.orc_filter_out_noise_gain(self.fout)

    def block(self):
        self.lib.orc_execute_filter_output(self.fout, self.rotate)
        y = self._out().copy()
        if self.dem is None:
            return y, None, None
        audio, sig = self.dem.block(y)
        return y, audio, sig


class Input:
    def __init__(self, pipe):
        self.lib = orc.lib(pipe.fast)
        self.fin = self.lib.orc_create_filter_input(pipe.L, pipe.M, orc.COMPLEX)
        self.lo2 = orc.new_osc(0.0, fast=pipe.fast)
        self.doppler = orc.new_osc(0.0, fast=pipe.fast)
        self.gain = 1.0
        self.n0 = float("nan")
        self.L = pipe.L
        self.N = pipe.L + pipe.M - 1

    def view(self, fn, count):
        p = fn(self.fin)
        buf = (C.c_char * (count * 8)).from_address(p)
        return np.frombuffer(buf, dtype=np.complex64, count=count)


class Pipeline:
    """One bank: n_inputs input streams, channels attached to them; the CPU twin of pyka9q.Bank."""

    def __init__(self, L, M, decimate, samprate, n_inputs=1, payload=0, compute_n0=False, n0_low=0.0, n0_high=0.0,
                 fast=False, threads=1):
        self.L, self.M, self.D, self.samprate = L, M, decimate, samprate
        self.threads = threads
        self.pool = None
        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=threads)
        self.N = L + M - 1
        self.fast = fast
        self.payload = payload
        self.compute_n0, self.n0_low, self.n0_high = compute_n0, n0_low, n0_high
        self.lib = orc.lib(fast)
        self.fins = [Input(self) for _ in range(n_inputs)]
        self.channels = []

    def add_channel(self, **kw):
        ch = Channel(self, **kw)
        self.channels.append(ch)
        return ch

    def set_lo(self, inp, lo2, doppler=0.0, doppler_rate=0.0):
        i = self.fins[inp]
        self.lib.orc_set_osc(C.addressof(i.lo2), lo2, 0.0)
        self.lib.orc_set_osc(C.addressof(i.doppler), doppler, doppler_rate)

    def convert(self, inp, raw):
        """radio.c:136-178: wire bytes -> cf32 (payload 0 = already cf32, scaled by gain_factor)."""
        i = self.fins[inp]
        if self.payload == 0:
            x = np.ascontiguousarray(raw, np.complex64)
            return (x * np.float32(i.gain)).astype(np.complex64) if i.gain != 1.0 else x
        raw = np.ascontiguousarray(raw).view(np.uint8)
        n = self.lib.orc_sampcount(self.payload, raw.size)
        out = np.empty(n, np.complex64)
        assert self.lib.orc_convert_samples(self.payload, raw.ctypes.data_as(C.c_void_p), raw.size, i.gain,
                                            out.ctypes.data_as(C.c_void_p)) == n
        return out

    def run_block(self, blocks):
        """blocks: one array of L raw samples per input.  Returns per-channel (baseband, audio, sig) and
        per-input (if_power, n0)."""
        istat = []
        for inp, raw in zip(self.fins, blocks):
            x = self.convert(self.fins.index(inp), raw)
            assert x.size == self.L
            mixed, energy = orc.mix(inp.doppler, inp.lo2, x, fast=self.fast)
            inp.view(self.lib.orc_filter_in_input, self.L)[:] = mixed
            self.lib.orc_execute_filter_input(inp.fin)
            if_power = energy / self.L  # radio.c:197
            n0 = float("nan")
            if self.compute_n0:
                fd = inp.view(self.lib.orc_filter_in_fdomain, self.N)
                v = self.lib.orc_compute_n0(fd.ctypes.data_as(C.c_void_p), self.N, self.n0_low, self.n0_high,
                                            self.samprate)
                if not np.isnan(inp.n0):  # radio.c:199-202
                    inp.n0 = float(np.float32(np.float64(np.float32(inp.n0)) + .005 * np.float64(np.float32(v) - np.float32(inp.n0))))
                else:
                    inp.n0 = v
                n0 = inp.n0
            istat.append((if_power, n0))
        if self.pool is None:
            outs = [ch.block() for ch in self.channels]
        else:  # channels partitioned across threads, one shared forward FFT per block (the C calls release the GIL)
            outs = list(self.pool.map(lambda ch: ch.block(), self.channels, chunksize=max(1, len(self.channels) // self.threads)))
        return outs, istat

    def spectrum(self, inp):
        i = self.fins[inp]
        return i.view(self.lib.orc_filter_in_fdomain, self.N).copy()
