"""ctypes view of libka9q_hip.so (include/ka9q_hip.h) for the tests and bench.py.

This is plumbing, not the product: every call goes straight to the C ABI that a C caller of the
reference's filter.h (filter.h:83-95) or of the bank extension would bind.  Loading fails loudly if the
shared library has not been built (`python __graft_entry__.py build`); there is no Python/CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KA9Q_HIP_LIB", os.path.join(_HERE, "libka9q_hip.so"))  # override: diagnostic builds only

NONE, COMPLEX, CROSS_CONJ, REAL = 0, 1, 2, 3
KH_DEMOD_NONE, KH_DEMOD_FM, KH_DEMOD_LINEAR, KH_DEMOD_AM = 0, 1, 2, 3
KH_PT_CF32, KH_PT_IQ12, KH_PT_IQ16LE, KH_PT_IQ8, KH_PT_PCM16BE = 0, 95, 97, 98, 10
KH_STAGES = ["ingest", "fft1", "fft2", "n0", "chan", "demod", "carry"]

# every symbol include/ka9q_hip.h declares
EXPORTS = [
    "ka9q_hip_init", "ka9q_hip_last_error", "ka9q_hip_alloc", "ka9q_hip_free",
    "create_filter_input", "create_filter_input_file", "notch_create", "notch", "kh_notch_block", "create_filter_output", "execute_filter_input", "execute_filter_output",
    "delete_filter_input", "delete_filter_output", "make_kaiser", "window_filter", "window_rfilter",
    "set_filter", "noise_gain", "hb15_block", "hb3_block", "kh_decimate_rails",
    "kh_bank_create", "kh_bank_destroy", "kh_bank_set_channel", "kh_bank_set_lo", "kh_bank_set_second_lo", "kh_bank_set_doppler", "kh_bank_write_input",
    "kh_bank_write_zeros", "kh_bank_set_gain", "kh_bank_raw_device_ptr", "kh_bank_commit_input",
    "kh_bank_debug_pending", "kh_bank_stream_open", "kh_bank_stream_slot", "kh_bank_stream_submit", "kh_bank_stream_close", "kh_bank_read_pcm_all",
    "kh_bank_execute", "kh_bank_sync", "kh_bank_read_baseband", "kh_bank_read_pcm", "kh_bank_read_audio",
    "kh_bank_read_status", "kh_bank_read_input_status", "kh_bank_read_spectrum", "kh_bank_read_response",
    "kh_bank_pcm_device_ptr", "kh_bank_keep_audio", "kh_bank_keep_baseband", "kh_bank_keep_spectrum", "kh_bank_channel_state_size", "kh_bank_get_channel_state", "kh_bank_set_channel_state", "kh_bank_use_graph", "kh_bank_stream", "kh_bank_profile", "kh_bank_stage_times",
    "kh_stage_kernel_name", "kh_bank_rtp_enable", "kh_bank_rtp_set_state", "kh_bank_read_rtp",
    "kh_afsk_create", "kh_afsk_destroy", "kh_afsk_write_pcm", "kh_afsk_pcm_device_ptr", "kh_afsk_commit_pcm",
    "kh_afsk_execute", "kh_afsk_sync", "kh_afsk_read_frames", "kh_afsk_read_bits", "kh_afsk_read_status",
    "kh_afsk_read_baseband", "kh_afsk_read_products",
    "kh_tones_create", "kh_tones_destroy", "kh_tones_write_pcm", "kh_tones_pcm_device_ptr", "kh_tones_commit_pcm",
    "kh_tones_execute", "kh_tones_sync", "kh_tones_read_pl", "kh_tones_read_dtmf",
    "kh_hackrf_correct", "kh_funcube_correct", "kh_hackrf_pack", "kh_hackrf_pack_noalloc",
]


class RC(C.Union):
    _fields_ = [("r", C.POINTER(C.c_float)), ("c", C.c_void_p)]


class PthreadMutex(C.Structure):  # glibc x86-64: 40 bytes, 8-byte aligned
    _fields_ = [("_", C.c_longlong * 5)]


class PthreadCond(C.Structure):  # glibc x86-64: 48 bytes, 8-byte aligned
    _fields_ = [("_", C.c_longlong * 6)]


class CF32(C.Structure):  # C's complex float, by value
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class NotchFilter(C.Structure):  # filter.h:97-103
    _fields_ = [("phase_re", C.c_double), ("phase_im", C.c_double), ("step_re", C.c_double), ("step_im", C.c_double),
                ("dc", CF32), ("bw", C.c_float)]


class FilterIn(C.Structure):  # filter.h:55-68
    _fields_ = [("in_type", C.c_int), ("ilen", C.c_uint), ("impulse_length", C.c_uint), ("fdomain", C.c_void_p),
                ("input_buffer", RC), ("input", RC), ("fwd_plan", C.c_void_p), ("blocknum", C.c_uint),
                ("filter_mutex", PthreadMutex), ("filter_cond", PthreadCond), ("fd", C.c_int)]


class FilterOut(C.Structure):  # filter.h:69-82
    _fields_ = [("master", C.POINTER(FilterIn)), ("out_type", C.c_int), ("response", C.c_void_p),
                ("response_mutex", PthreadMutex), ("f_fdomain", C.c_void_p), ("noise_gain", C.c_float),
                ("output_buffer", RC), ("output", RC), ("rev_plan", C.c_void_p), ("decimate", C.c_uint),
                ("olen", C.c_uint), ("blocknum", C.c_uint)]


class Hb15State(C.Structure):  # decimate.h:4-9
    _fields_ = [("coeffs", C.c_float * 4), ("even_samples", C.c_float * 4), ("odd_samples", C.c_float * 4),
                ("old_odd_samples", C.c_float * 4)]


class BankConfig(C.Structure):
    _fields_ = [("L", C.c_uint), ("M", C.c_uint), ("decimate", C.c_uint), ("n_inputs", C.c_uint),
                ("n_channels", C.c_uint), ("max_blocks", C.c_uint), ("samprate", C.c_int), ("payload", C.c_int),
                ("pcm_big_endian", C.c_int), ("compute_n0", C.c_int), ("n0_low", C.c_float), ("n0_high", C.c_float)]


class ChannelConfig(C.Structure):
    _fields_ = [("input", C.c_uint), ("rotate", C.c_int), ("out_type", C.c_int), ("demod", C.c_int),
                ("low", C.c_float), ("high", C.c_float), ("kaiser_beta", C.c_float), ("flat", C.c_int),
                ("headroom", C.c_float), ("pll", C.c_int), ("square", C.c_int), ("agc", C.c_int), ("env", C.c_int),
                ("channels", C.c_int), ("hangtime", C.c_float), ("recovery_rate", C.c_float),
                ("agc_gain", C.c_float), ("loop_bw", C.c_float), ("shift", C.c_double)]


class Status(C.Structure):
    _fields_ = [("bb_power", C.c_float), ("snr", C.c_float), ("foffset", C.c_float), ("pdeviation", C.c_float),
                ("cphase", C.c_float), ("lock_timer", C.c_float), ("pll_lock", C.c_int),
                ("output_level", C.c_float), ("agc_gain", C.c_float), ("squelch_open", C.c_int),
                ("reserved", C.c_float * 2)]


STATUS_DTYPE = np.dtype([("bb_power", "f4"), ("snr", "f4"), ("foffset", "f4"), ("pdeviation", "f4"),
                         ("cphase", "f4"), ("lock_timer", "f4"), ("pll_lock", "i4"), ("output_level", "f4"),
                         ("agc_gain", "f4"), ("squelch_open", "i4"), ("reserved", "f4", (2,))])
INPUT_STATUS_DTYPE = np.dtype([("if_power", "f4"), ("n0", "f4")])

_lib = None


def lib():
    """Load libka9q_hip.so (once).  Raises if it is missing: the product has no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build`")
    L = C.CDLL(LIB_PATH)
    L.ka9q_hip_init.argtypes = [C.c_int]
    L.ka9q_hip_last_error.restype = C.c_char_p
    L.ka9q_hip_alloc.restype = C.c_void_p
    L.ka9q_hip_alloc.argtypes = [C.c_size_t]
    L.ka9q_hip_free.argtypes = [C.c_void_p]
    L.create_filter_input.restype = C.POINTER(FilterIn)
    L.create_filter_input.argtypes = [C.c_uint, C.c_uint, C.c_int]
    L.create_filter_input_file.restype = C.POINTER(FilterIn)
    L.create_filter_input_file.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_char_p]
    L.notch_create.restype = C.POINTER(NotchFilter)
    L.notch_create.argtypes = [C.c_double, C.c_float]
    L.notch.restype = CF32
    L.notch.argtypes = [C.POINTER(NotchFilter), CF32]
    L.kh_notch_block.argtypes = [C.POINTER(NotchFilter), C.c_void_p, C.c_int]
    L.create_filter_output.restype = C.POINTER(FilterOut)
    L.create_filter_output.argtypes = [C.POINTER(FilterIn), C.c_void_p, C.c_uint, C.c_int]
    L.execute_filter_input.argtypes = [C.POINTER(FilterIn)]
    L.execute_filter_output.argtypes = [C.POINTER(FilterOut), C.c_int]
    L.delete_filter_input.argtypes = [C.POINTER(FilterIn)]
    L.delete_filter_output.argtypes = [C.POINTER(FilterOut)]
    L.make_kaiser.argtypes = [C.c_void_p, C.c_uint, C.c_float]
    L.window_filter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float]
    L.window_rfilter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float]
    L.set_filter.argtypes = [C.POINTER(FilterOut), C.c_float, C.c_float, C.c_float]
    L.noise_gain.restype = C.c_float
    L.noise_gain.argtypes = [C.POINTER(FilterOut)]
    L.hb15_block.restype = None
    L.hb15_block.argtypes = [C.POINTER(Hb15State), C.c_void_p, C.c_void_p, C.c_int]
    L.hb3_block.restype = None
    L.hb3_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.kh_decimate_rails.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                    C.c_int]
    L.kh_bank_create.restype = C.c_void_p
    L.kh_bank_create.argtypes = [C.POINTER(BankConfig)]
    L.kh_bank_destroy.argtypes = [C.c_void_p]
    L.kh_bank_set_channel.argtypes = [C.c_void_p, C.c_uint, C.POINTER(ChannelConfig)]
    L.kh_bank_set_lo.argtypes = [C.c_void_p, C.c_uint, C.c_double, C.c_double, C.c_double]
    L.kh_bank_set_second_lo.argtypes = [C.c_void_p, C.c_uint, C.c_double]
    L.kh_bank_set_doppler.argtypes = [C.c_void_p, C.c_uint, C.c_double, C.c_double]
    L.kh_bank_set_gain.argtypes = [C.c_void_p, C.c_uint, C.c_float]
    L.kh_bank_write_input.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_bank_write_zeros.argtypes = [C.c_void_p, C.c_uint, C.c_size_t]
    L.kh_bank_raw_device_ptr.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.kh_bank_commit_input.argtypes = [C.c_void_p, C.c_uint, C.c_size_t]
    L.kh_bank_execute.argtypes = [C.c_void_p, C.c_uint]
    L.kh_bank_sync.argtypes = [C.c_void_p]
    L.kh_bank_debug_pending.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.kh_bank_stream_open.argtypes = [C.c_void_p, C.c_uint]
    L.kh_bank_stream_slot.restype = C.c_void_p
    L.kh_bank_stream_slot.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_size_t)]
    L.kh_bank_stream_submit.argtypes = [C.c_void_p, C.c_uint]
    L.kh_bank_stream_close.argtypes = [C.c_void_p]
    L.kh_bank_read_pcm_all.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for name in ("kh_bank_read_baseband", "kh_bank_read_pcm", "kh_bank_read_audio", "kh_bank_read_status",
                 "kh_bank_read_input_status"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_bank_read_spectrum.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    L.kh_bank_read_response.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.POINTER(C.c_float)]
    L.kh_bank_pcm_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.kh_bank_stream.restype = C.c_void_p
    L.kh_bank_stream.argtypes = [C.c_void_p]
    L.kh_bank_profile.argtypes = [C.c_void_p, C.c_int]
    L.kh_bank_keep_audio.argtypes = [C.c_void_p, C.c_int]
    L.kh_bank_rtp_enable.argtypes = [C.c_void_p, C.c_int]
    L.kh_bank_rtp_set_state.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    L.kh_bank_read_rtp.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    L.kh_bank_keep_baseband.argtypes = [C.c_void_p, C.c_int]
    L.kh_bank_keep_spectrum.argtypes = [C.c_void_p, C.c_int]
    L.kh_bank_channel_state_size.argtypes = []
    L.kh_bank_channel_state_size.restype = C.c_size_t
    L.kh_bank_get_channel_state.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_bank_set_channel_state.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_bank_use_graph.argtypes = [C.c_void_p, C.c_int]
    L.kh_bank_stage_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
    L.kh_stage_kernel_name.restype = C.c_char_p
    L.kh_stage_kernel_name.argtypes = [C.c_int]
    L.kh_afsk_create.restype = C.c_void_p
    L.kh_afsk_create.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.kh_afsk_destroy.argtypes = [C.c_void_p]
    L.kh_afsk_write_pcm.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_afsk_pcm_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.kh_afsk_commit_pcm.argtypes = [C.c_void_p, C.c_uint, C.c_size_t]
    L.kh_afsk_execute.argtypes = [C.c_void_p, C.c_uint]
    L.kh_afsk_sync.argtypes = [C.c_void_p]
    L.kh_afsk_read_frames.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.kh_afsk_read_bits.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.kh_afsk_read_status.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    L.kh_afsk_read_baseband.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_afsk_read_products.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_tones_create.restype = C.c_void_p
    L.kh_tones_create.argtypes = [C.c_uint, C.c_uint]
    L.kh_tones_destroy.argtypes = [C.c_void_p]
    L.kh_tones_write_pcm.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t]
    L.kh_tones_pcm_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.kh_tones_commit_pcm.argtypes = [C.c_void_p, C.c_uint, C.c_size_t]
    L.kh_tones_execute.argtypes = [C.c_void_p, C.c_uint]
    L.kh_tones_sync.argtypes = [C.c_void_p]
    L.kh_tones_read_pl.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.kh_tones_read_dtmf.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.kh_hackrf_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.kh_funcube_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_float]
    L.kh_hackrf_pack.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_float, C.c_void_p, C.POINTER(C.c_float)]
    _lib = L
    return L


def last_error():
    return lib().ka9q_hip_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {last_error()}")


def _as_np(ptr, count, dtype):
    """numpy view of `count` items at a host address returned by the library (pinned memory)."""
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


class Filter:
    """One filter_in + one filter_out driven exactly as radio.c/fm.c drive them (single threaded here)."""

    def __init__(self, L, M, decimate, in_type=COMPLEX, out_type=COMPLEX):
        self.lib = lib()
        self.L, self.M, self.D = L, M, decimate
        self.N = L + M - 1
        self.fin = self.lib.create_filter_input(L, M, in_type)
        if not self.fin:
            raise RuntimeError("create_filter_input: " + last_error())
        self.fout = self.lib.create_filter_output(self.fin, None, decimate, out_type)
        if not self.fout:
            err = last_error()
            self.lib.delete_filter_input(self.fin)
            self.fin = None
            raise RuntimeError("create_filter_output: " + err)
        self.in_type, self.out_type = in_type, out_type
        self.olen = self.fout.contents.olen

    def input(self):
        if self.in_type == REAL:
            return _as_np(C.cast(self.fin.contents.input.r, C.c_void_p).value, self.L, np.float32)
        return _as_np(self.fin.contents.input.c, self.L, np.complex64)

    def fdomain(self):
        n = self.N // 2 + 1 if self.in_type == REAL else self.N
        return _as_np(self.fin.contents.fdomain, n, np.complex64)

    def output(self):
        if self.fout.contents.out_type == REAL:
            return _as_np(C.cast(self.fout.contents.output.r, C.c_void_p).value, self.olen, np.float32)
        return _as_np(self.fout.contents.output.c, self.olen, np.complex64)

    def response(self):
        nd = self.N // self.D
        n = nd // 2 + 1 if (self.in_type == REAL and self.out_type == REAL) else nd
        return _as_np(self.fout.contents.response, n, np.complex64)

    def set_response(self, resp):
        """Hand a caller-designed response to the filter (modulate.c:121-138 pattern)."""
        resp = np.ascontiguousarray(resp, dtype=np.complex64)
        p = self.lib.ka9q_hip_alloc(resp.nbytes)
        C.memmove(p, resp.ctypes.data, resp.nbytes)
        old = self.fout.contents.response
        self.fout.contents.response = p
        if old:
            self.lib.ka9q_hip_free(old)

    def set_filter(self, low, high, beta):
        _check(self.lib.set_filter(self.fout, low, high, beta), "set_filter")

    def noise_gain(self):
        return self.lib.noise_gain(self.fout)

    def set_out_type(self, t):  # radio.c:194
        self.fout.contents.out_type = t

    def execute(self, block, rotate=0):
        self.input()[:] = block
        _check(self.lib.execute_filter_input(self.fin), "execute_filter_input")
        _check(self.lib.execute_filter_output(self.fout, rotate), "execute_filter_output")
        return self.output().copy()

    def close(self):
        if self.fout:
            self.lib.delete_filter_output(self.fout)
            self.fout = None
        if self.fin:
            self.lib.delete_filter_input(self.fin)
            self.fin = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bank:
    def __init__(self, L, M, decimate, n_inputs, n_channels, max_blocks, samprate, payload=KH_PT_CF32,
                 pcm_big_endian=0, compute_n0=0, n0_low=0.0, n0_high=0.0):
        self.lib = lib()
        self.cfg = BankConfig(L, M, decimate, n_inputs, n_channels, max_blocks, samprate, payload, pcm_big_endian,
                              compute_n0, n0_low, n0_high)
        self.h = self.lib.kh_bank_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("kh_bank_create: " + last_error())
        self.L, self.M, self.D = L, M, decimate
        self.N = L + M - 1
        self.ND = self.N // decimate
        self.olen = L // decimate
        self.T = max_blocks
        self.C = n_channels
        self.P = n_inputs
        self.nch = {}

    def set_channel(self, ch, **kw):
        d = dict(input=0, rotate=0, out_type=COMPLEX, demod=KH_DEMOD_NONE, low=-5000.0, high=5000.0,
                 kaiser_beta=3.0, flat=0, headroom=10 ** (-15 / 20), pll=0, square=0, agc=0, env=0, channels=1,
                 hangtime=0.0, recovery_rate=1.0, agc_gain=1.0, loop_bw=5.0, shift=0.0)
        d.update(kw)
        cc = ChannelConfig(**d)
        _check(self.lib.kh_bank_set_channel(self.h, ch, C.byref(cc)), "kh_bank_set_channel")
        self.nch[ch] = d["channels"]

    def set_lo(self, inp, lo2, doppler=0.0, doppler_rate=0.0):
        _check(self.lib.kh_bank_set_lo(self.h, inp, lo2, doppler, doppler_rate), "kh_bank_set_lo")

    def set_second_lo(self, inp, lo2):
        _check(self.lib.kh_bank_set_second_lo(self.h, inp, lo2), "kh_bank_set_second_lo")

    def set_doppler(self, inp, f, rate=0.0):
        _check(self.lib.kh_bank_set_doppler(self.h, inp, f, rate), "kh_bank_set_doppler")

    def set_gain(self, inp, g):
        _check(self.lib.kh_bank_set_gain(self.h, inp, g), "kh_bank_set_gain")

    def write_input(self, inp, arr, count=None):
        arr = np.ascontiguousarray(arr)
        if count is None:
            count = arr.shape[0] if self.cfg.payload != KH_PT_IQ12 else arr.nbytes // 3
        _check(self.lib.kh_bank_write_input(self.h, inp, arr.ctypes.data, count), "kh_bank_write_input")

    def write_zeros(self, inp, count):
        _check(self.lib.kh_bank_write_zeros(self.h, inp, count), "kh_bank_write_zeros")

    def raw_device_ptr(self, inp):
        p, cap = C.c_void_p(), C.c_size_t()
        _check(self.lib.kh_bank_raw_device_ptr(self.h, inp, C.byref(p), C.byref(cap)), "kh_bank_raw_device_ptr")
        return p.value, cap.value

    def commit_input(self, inp, count):
        _check(self.lib.kh_bank_commit_input(self.h, inp, count), "kh_bank_commit_input")

    def execute(self, nblocks, sync=True):
        _check(self.lib.kh_bank_execute(self.h, nblocks), "kh_bank_execute")
        if sync:
            self.sync()

    def sync(self):
        _check(self.lib.kh_bank_sync(self.h), "kh_bank_sync")

    def baseband(self, ch, nblocks):
        out = np.empty(nblocks * self.olen, np.complex64)
        _check(self.lib.kh_bank_read_baseband(self.h, ch, out.ctypes.data, out.size), "kh_bank_read_baseband")
        return out

    def pcm(self, ch, nblocks):
        out = np.empty(nblocks * self.olen * self.nch.get(ch, 1), np.int16)
        _check(self.lib.kh_bank_read_pcm(self.h, ch, out.ctypes.data, out.size), "kh_bank_read_pcm")
        return out

    def pcm_all(self, nblocks, out=None):
        """PCM of every channel (mono banks) in one transfer: array [C, nblocks*olen]"""
        nw = nblocks * self.olen
        if out is None:
            out = np.empty((self.C, nw), np.int16)
        _check(self.lib.kh_bank_read_pcm_all(self.h, out.ctypes.data, nw), "kh_bank_read_pcm_all")
        return out

    def debug_pending(self):
        buf = C.create_string_buffer(512)
        self.lib.kh_bank_debug_pending(self.h, buf, 512)
        return buf.value.decode()

    def stream_open(self, depth):
        _check(self.lib.kh_bank_stream_open(self.h, depth), "kh_bank_stream_open")

    def stream_slot(self, inp, dtype, count):
        """numpy view of the first `count` items of the pinned slot the next submit will transfer for input `inp`"""
        cap = C.c_size_t()
        p = self.lib.kh_bank_stream_slot(self.h, inp, C.byref(cap))
        if not p:
            raise RuntimeError("kh_bank_stream_slot: " + last_error())
        return _as_np(p, count, dtype)

    def stream_submit(self, nblocks):
        _check(self.lib.kh_bank_stream_submit(self.h, nblocks), "kh_bank_stream_submit")

    def stream_close(self):
        _check(self.lib.kh_bank_stream_close(self.h), "kh_bank_stream_close")

    def audio(self, ch, nblocks):
        out = np.empty(nblocks * self.olen * self.nch.get(ch, 1), np.float32)
        _check(self.lib.kh_bank_read_audio(self.h, ch, out.ctypes.data, out.size), "kh_bank_read_audio")
        return out

    def status(self, ch, nblocks):
        out = np.zeros(nblocks, STATUS_DTYPE)
        assert STATUS_DTYPE.itemsize == C.sizeof(Status)
        _check(self.lib.kh_bank_read_status(self.h, ch, out.ctypes.data, nblocks), "kh_bank_read_status")
        return out

    def input_status(self, inp, nblocks):
        out = np.zeros(nblocks, INPUT_STATUS_DTYPE)
        _check(self.lib.kh_bank_read_input_status(self.h, inp, out.ctypes.data, nblocks),
               "kh_bank_read_input_status")
        return out

    def spectrum(self, inp, block):
        out = np.empty(self.N, np.complex64)
        _check(self.lib.kh_bank_read_spectrum(self.h, inp, block, out.ctypes.data), "kh_bank_read_spectrum")
        return out

    def response(self, ch):
        out = np.empty(self.ND, np.complex64)
        ng = C.c_float()
        _check(self.lib.kh_bank_read_response(self.h, ch, out.ctypes.data, C.byref(ng)), "kh_bank_read_response")
        return out, ng.value

    RTP_CHUNK = np.dtype([("timestamp", "<u4"), ("seq", "<u2"), ("marker", "u1"), ("send", "u1"), ("offset", "<u4"),
                          ("nwords", "<u2"), ("reserved", "<u2")])
    RTP_STATE = np.dtype([("timestamp", "<u4"), ("seq", "<u2"), ("silent", "<u2"), ("packets", "<u8"), ("bytes", "<u8"),
                          ("samples", "<u8")])

    def rtp_enable(self, enable=1):
        _check(self.lib.kh_bank_rtp_enable(self.h, int(enable)), "kh_bank_rtp_enable")

    def rtp_set_state(self, ch, timestamp=0, seq=0, silent=0):
        s = np.zeros(1, self.RTP_STATE)
        s["timestamp"], s["seq"], s["silent"] = timestamp, seq, silent
        _check(self.lib.kh_bank_rtp_set_state(self.h, ch, s.ctypes.data), "kh_bank_rtp_set_state")

    def rtp(self, ch):
        n = C.c_size_t()
        _check(self.lib.kh_bank_read_rtp(self.h, ch, None, 0, C.byref(n), None), "kh_bank_read_rtp")
        out = np.zeros(n.value, self.RTP_CHUNK)
        after = np.zeros(1, self.RTP_STATE)
        _check(self.lib.kh_bank_read_rtp(self.h, ch, out.ctypes.data, n.value, C.byref(n), after.ctypes.data), "kh_bank_read_rtp")
        return out, after[0]

    def keep_audio(self, enable):
        _check(self.lib.kh_bank_keep_audio(self.h, int(enable)), "kh_bank_keep_audio")

    def use_graph(self, enable):
        _check(self.lib.kh_bank_use_graph(self.h, int(enable)), "kh_bank_use_graph")

    def get_channel_state(self, ch):
        buf = C.create_string_buffer(self.lib.kh_bank_channel_state_size())
        _check(self.lib.kh_bank_get_channel_state(self.h, ch, buf, len(buf)), "kh_bank_get_channel_state")
        return buf.raw

    def set_channel_state(self, ch, blob):
        _check(self.lib.kh_bank_set_channel_state(self.h, ch, blob, len(blob)), "kh_bank_set_channel_state")

    def keep_baseband(self, enable):
        _check(self.lib.kh_bank_keep_baseband(self.h, int(enable)), "kh_bank_keep_baseband")

    def keep_spectrum(self, enable):
        _check(self.lib.kh_bank_keep_spectrum(self.h, int(enable)), "kh_bank_keep_spectrum")

    def write_input_device(self, inp, dptr, count):
        _check(self.lib.kh_bank_write_input(self.h, inp, dptr, count), "kh_bank_write_input")

    def profile(self, enable):
        _check(self.lib.kh_bank_profile(self.h, int(enable)), "kh_bank_profile")

    def stage_times(self):
        ms = (C.c_float * len(KH_STAGES))()
        n = (C.c_uint * len(KH_STAGES))()
        _check(self.lib.kh_bank_stage_times(self.h, ms, n), "kh_bank_stage_times")
        return {KH_STAGES[i]: (ms[i], n[i]) for i in range(len(KH_STAGES))}

    def close(self):
        if self.h:
            self.lib.kh_bank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Afsk:
    """packet.c's receive side for `sessions` PCM streams (kh_afsk_*): AFSK-1200 + HDLC + CRC on the device."""
    L, M = 1000, 1049
    FRAME = np.dtype([("sample", "<u8"), ("timestamp", "<u4"), ("seq", "<u2"), ("bytes", "<u2"), ("data", "u1", (1024,))])
    BIT = np.dtype([("sample", "<u8"), ("cur_val", "<f4"), ("reserved", "<u4")])
    STATUS = np.dtype([("timestamp", "<u4"), ("seq", "<u2"), ("flagsync", "<u2"), ("packets", "<u8"), ("bytes", "<u8"),
                       ("frames_last_launch", "<u4"), ("bits_last_launch", "<u4")])

    def __init__(self, sessions, max_blocks, max_frames=16, max_bits=0):
        self.lib = lib()
        self.S, self.T, self.max_frames, self.max_bits = sessions, max_blocks, max_frames, max_bits
        self.h = self.lib.kh_afsk_create(sessions, max_blocks, max_frames, max_bits)
        if not self.h:
            raise RuntimeError("kh_afsk_create: " + last_error())

    def write_pcm(self, session, pcm_be):
        pcm_be = np.ascontiguousarray(pcm_be, np.int16)
        _check(self.lib.kh_afsk_write_pcm(self.h, session, pcm_be.ctypes.data, pcm_be.size), "kh_afsk_write_pcm")

    def pcm_device_ptr(self):
        p, st = C.c_void_p(), C.c_size_t()
        _check(self.lib.kh_afsk_pcm_device_ptr(self.h, C.byref(p), C.byref(st)), "kh_afsk_pcm_device_ptr")
        return p.value, st.value

    def commit_pcm(self, session, nwords):
        _check(self.lib.kh_afsk_commit_pcm(self.h, session, nwords), "kh_afsk_commit_pcm")

    def execute(self, nblocks, sync=True):
        _check(self.lib.kh_afsk_execute(self.h, nblocks), "kh_afsk_execute")
        if sync:
            self.sync()

    def sync(self):
        _check(self.lib.kh_afsk_sync(self.h), "kh_afsk_sync")

    def frames(self, session):
        out = np.zeros(self.max_frames, self.FRAME)
        n = C.c_size_t()
        _check(self.lib.kh_afsk_read_frames(self.h, session, out.ctypes.data, self.max_frames, C.byref(n)), "kh_afsk_read_frames")
        return [dict(sample=int(f["sample"]), seq=int(f["seq"]), timestamp=int(f["timestamp"]),
                     data=bytes(f["data"][:f["bytes"]])) for f in out[:n.value]]

    def bits(self, session):
        out = np.zeros(max(self.max_bits, 1), self.BIT)
        n = C.c_size_t()
        _check(self.lib.kh_afsk_read_bits(self.h, session, out.ctypes.data, self.max_bits, C.byref(n)), "kh_afsk_read_bits")
        return out[:n.value]

    def status(self, session):
        out = np.zeros(1, self.STATUS)
        _check(self.lib.kh_afsk_read_status(self.h, session, out.ctypes.data), "kh_afsk_read_status")
        return out[0]

    def baseband(self, session, nblocks):
        out = np.empty(nblocks * self.L, np.complex64)
        _check(self.lib.kh_afsk_read_baseband(self.h, session, out.ctypes.data, out.size), "kh_afsk_read_baseband")
        return out

    def products(self, session, nblocks):
        """tone products of the last launch [nblocks*L, 4] (kh_afsk_mix): what the decoder's accumulators add up"""
        out = np.empty((nblocks * self.L, 4), np.float32)
        _check(self.lib.kh_afsk_read_products(self.h, session, out.ctypes.data, nblocks * self.L), "kh_afsk_read_products")
        return out

    def close(self):
        if self.h:
            self.lib.kh_afsk_destroy(self.h)
            self.h = None


class Tones:
    """pl.c's CTCSS and DTMF decoders for `sessions` PCM streams (kh_tones_*)."""
    WINDOW = 2400
    PL = np.dtype([("tone", "<f4"), ("snr", "<f4"), ("e0", "<f4"), ("e1", "<f4"), ("index", "<i4")])
    DTMF = np.dtype([("digit", "<i4"), ("low_index", "<i4"), ("high_index", "<i4"), ("low_snr", "<f4"), ("high_snr", "<f4")])

    def __init__(self, sessions, max_windows):
        self.lib = lib()
        self.S, self.W = sessions, max_windows
        self.h = self.lib.kh_tones_create(sessions, max_windows)
        if not self.h:
            raise RuntimeError("kh_tones_create: " + last_error())

    def write_pcm(self, session, pcm_be):
        pcm_be = np.ascontiguousarray(pcm_be, np.int16)
        _check(self.lib.kh_tones_write_pcm(self.h, session, pcm_be.ctypes.data, pcm_be.size), "kh_tones_write_pcm")

    def commit_pcm(self, session, nwords):
        _check(self.lib.kh_tones_commit_pcm(self.h, session, nwords), "kh_tones_commit_pcm")

    def execute(self, nwindows, sync=True):
        _check(self.lib.kh_tones_execute(self.h, nwindows), "kh_tones_execute")
        if sync:
            _check(self.lib.kh_tones_sync(self.h), "kh_tones_sync")

    def sync(self):
        _check(self.lib.kh_tones_sync(self.h), "kh_tones_sync")

    def pl(self, session):
        out = np.zeros(self.W // 5 + 1, self.PL)
        n = C.c_size_t()
        _check(self.lib.kh_tones_read_pl(self.h, session, out.ctypes.data, out.size, C.byref(n)), "kh_tones_read_pl")
        return out[:n.value]

    def dtmf(self, session):
        out = np.zeros(self.W, self.DTMF)
        n = C.c_size_t()
        _check(self.lib.kh_tones_read_dtmf(self.h, session, out.ctypes.data, out.size, C.byref(n)), "kh_tones_read_dtmf")
        return out[:n.value]

    def close(self):
        if self.h:
            self.lib.kh_tones_destroy(self.h)
            self.h = None


FE_STATE = np.dtype([("dc_re", "<f4"), ("dc_im", "<f4"), ("sinphi", "<f4"), ("imbalance", "<f4"), ("gain_i", "<f4"),
                     ("gain_q", "<f4"), ("secphi", "<f4"), ("tanphi", "<f4"), ("rotate_phase", "<i4"), ("clips", "<i4"),
                     ("in_power", "<f4")])


def fe_state(imbalance=0.0):
    s = np.zeros(1, FE_STATE)
    s["imbalance"] = imbalance
    s["gain_i"] = s["gain_q"] = s["secphi"] = 1.0
    return s


def hackrf_correct(state, iq, samples, nblocks, offset, rate_factor):
    iq = np.ascontiguousarray(iq, np.int8)
    assert iq.size == 2 * samples * nblocks
    oi, oq = np.empty(samples * nblocks, np.float32), np.empty(samples * nblocks, np.float32)
    _check(lib().kh_hackrf_correct(state.ctypes.data, iq.ctypes.data, samples, nblocks, offset, rate_factor, oi.ctypes.data,
                                   oq.ctypes.data), "kh_hackrf_correct")
    return oi, oq


def funcube_correct(state, sampbuf, blocksize, nblocks, rate_factor):
    buf = np.ascontiguousarray(sampbuf, np.int16).copy()
    assert buf.size == 2 * blocksize * nblocks
    _check(lib().kh_funcube_correct(state.ctypes.data, buf.ctypes.data, blocksize, nblocks, rate_factor), "kh_funcube_correct")
    return buf


def hackrf_pack(payload, ip, qp, atten):
    ip, qp = np.ascontiguousarray(ip, np.float32), np.ascontiguousarray(qp, np.float32)
    out = np.zeros(ip.size * {KH_PT_IQ16LE: 4, KH_PT_IQ12: 3, KH_PT_IQ8: 2}[payload], np.uint8)
    e = C.c_float()
    _check(lib().kh_hackrf_pack(payload, ip.ctypes.data, qp.ctypes.data, ip.size, atten, out.ctypes.data, C.byref(e)), "kh_hackrf_pack")
    return out, e.value
