"""ctypes view of the CPU oracle (oracle/ka9q_oracle.h) and of the compiled reference objects (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under ka9q-radio_amd/ may import this.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KA9Q_ORACLE_SO: another build of the same source, e.g. the sanitizer build of `make -C oracle asan`
ORACLE_SO = os.environ.get("KA9Q_ORACLE_SO") or os.path.join(_HERE, "libka9q_oracle.so")
ORACLE_FAST_SO = os.path.join(_HERE, "libka9q_oracle_fast.so")
REF_SO = os.path.join(_HERE, "_ref", "libka9q_ref.so")
REF_PORTABLE_SO = os.path.join(_HERE, "_ref", "libka9q_ref_portable.so")

NONE, COMPLEX, CROSS_CONJ, REAL = 0, 1, 2, 3
IQ_PT12, IQ_PT, IQ_PT8, PCM_STEREO_PT = 95, 97, 98, 10


class Cd(C.Structure):  # double complex
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class Cf(C.Structure):  # float complex
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class Osc(C.Structure):  # struct orc_osc
    _fields_ = [("freq", C.c_double), ("rate", C.c_double), ("phasor", Cd), ("phasor_step", Cd),
                ("phasor_step_step", Cd), ("steps", C.c_int)]


class RefOsc(C.Structure):  # the reference's struct osc (osc.h:9-17) incl. its pthread_mutex_t (40 bytes)
    _fields_ = [("freq", C.c_double), ("rate", C.c_double), ("phasor", Cd), ("phasor_step", Cd),
                ("phasor_step_step", Cd), ("mutex", C.c_longlong * 5), ("steps", C.c_int)]


class Pll(C.Structure):  # struct orc_pll
    _fields_ = [("samptime", C.c_float), ("vco", Osc), ("integrator_gain", C.c_float), ("prop_gain", C.c_float),
                ("integrator", C.c_float)]


class RefPll(C.Structure):  # osc.h:19-25
    _fields_ = [("samptime", C.c_float), ("vco", RefOsc), ("integrator_gain", C.c_float), ("prop_gain", C.c_float),
                ("integrator", C.c_float)]


class Hb15(C.Structure):  # decimate.h:4-9
    _fields_ = [("coeffs", C.c_float * 4), ("even_samples", C.c_float * 4), ("odd_samples", C.c_float * 4),
                ("old_odd_samples", C.c_float * 4)]


class FmParams(C.Structure):
    _fields_ = [("input_samprate", C.c_int), ("decimate", C.c_int), ("headroom", C.c_float), ("low", C.c_float),
                ("high", C.c_float), ("flat", C.c_int)]


class FmState(C.Structure):
    _fields_ = [("state", Cf), ("lastaudio", C.c_float), ("squelch_open", C.c_int)]


class Sig(C.Structure):
    _fields_ = [("bb_power", C.c_float), ("snr", C.c_float), ("foffset", C.c_float), ("pdeviation", C.c_float),
                ("cphase", C.c_float), ("lock_timer", C.c_float), ("pll_lock", C.c_int), ("output_level", C.c_float),
                ("agc_gain", C.c_float), ("if_power", C.c_float), ("n0", C.c_float)]


class LinearParams(C.Structure):
    _fields_ = [("output_samprate", C.c_int), ("pll", C.c_int), ("square", C.c_int), ("agc", C.c_int),
                ("env", C.c_int), ("channels", C.c_int), ("headroom", C.c_float), ("hangtime", C.c_float),
                ("recovery_rate", C.c_float), ("loop_bw", C.c_float)]


class LinearState(C.Structure):
    _fields_ = [("hangcount", C.c_int), ("lock_count", C.c_int), ("pll", Pll), ("shift", Osc),
                ("agc_gain", C.c_float), ("snr", C.c_float), ("pll_lock", C.c_int)]


class AmParams(C.Structure):
    _fields_ = [("input_samprate", C.c_int), ("decimate", C.c_int), ("output_samprate", C.c_int),
                ("headroom", C.c_float), ("hangtime", C.c_float), ("recovery_rate", C.c_float)]


class AmState(C.Structure):
    _fields_ = [("hangcount", C.c_int), ("recovery_rate_per_sample", C.c_float), ("prev_recovery_rate", C.c_float),
                ("DC_filter", C.c_float), ("agc_gain", C.c_float)]


class PcmState(C.Structure):
    _fields_ = [("timestamp", C.c_uint32), ("seq", C.c_uint16), ("silent", C.c_int), ("packets", C.c_longlong),
                ("bytes", C.c_longlong), ("samples", C.c_longlong)]


class AfskState(C.Structure):  # struct orc_afsk_state
    _fields_ = [("mark", Osc), ("space", Osc), ("symphase", C.c_int), ("mark_accum", Cf), ("space_accum", Cf),
                ("mark_offset_accum", Cf), ("space_offset_accum", Cf), ("last_val", C.c_float), ("mid_val", C.c_float),
                ("hdlc_frame", C.c_ubyte * 1024), ("frame_bit", C.c_int), ("flagsync", C.c_int), ("ones", C.c_int),
                ("out_timestamp", C.c_uint32), ("out_seq", C.c_uint16), ("out_packets", C.c_longlong),
                ("out_bytes", C.c_longlong)]


class Ax25Frame(C.Structure):  # struct orc_ax25_frame
    _fields_ = [("sample", C.c_int), ("bytes", C.c_int), ("timestamp", C.c_uint32), ("seq", C.c_uint16),
                ("data", C.c_ubyte * 1024)]


class AfskBit(C.Structure):  # struct orc_afsk_bit
    _fields_ = [("sample", C.c_int), ("cur_val", C.c_float)]


class PlSession(C.Structure):  # struct orc_pl_session
    _fields_ = [("pl_integrators", Cf * 55), ("pl_osc", Osc * 55), ("dtmf_low_integrators", Cf * 4),
                ("dtmf_high_integrators", Cf * 4), ("dtmf_low_osc", Osc * 4), ("dtmf_high_osc", Osc * 4),
                ("pl_audio_count", C.c_int), ("dtmf_audio_count", C.c_int)]


class PlResult(C.Structure):
    _fields_ = [("tone", C.c_float), ("snr", C.c_float), ("e0", C.c_float), ("e1", C.c_float), ("index", C.c_int)]


class DtmfResult(C.Structure):
    _fields_ = [("digit", C.c_int), ("low_index", C.c_int), ("high_index", C.c_int), ("low_snr", C.c_float),
                ("high_snr", C.c_float)]


class FeState(C.Structure):  # struct orc_fe_state
    _fields_ = [("DC", Cf), ("sinphi", C.c_float), ("imbalance", C.c_float), ("gain_i", C.c_float), ("gain_q", C.c_float),
                ("secphi", C.c_float), ("tanphi", C.c_float), ("rotate_phase", C.c_int), ("clips", C.c_int),
                ("in_power", C.c_float)]


class Notch(C.Structure):
    _fields_ = [("osc_phase", Cd), ("osc_step", Cd), ("dcstate", Cf), ("bw", C.c_float)]


_libs = {}


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def lib(fast=False):
    path = ORACLE_FAST_SO if fast else ORACLE_SO
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle` (or `python __graft_entry__.py build`)")
    L = C.CDLL(path)
    vp, i, u, f, d = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_double
    L.orc_fft.argtypes = [i, vp, vp, i]
    L.orc_fft.restype = None
    L.orc_fft_r2c.argtypes = [i, vp, vp]
    L.orc_fft_c2r.argtypes = [i, vp, vp]
    L.orc_create_filter_input.restype = vp
    L.orc_create_filter_input.argtypes = [u, u, i]
    L.orc_create_filter_output.restype = vp
    L.orc_create_filter_output.argtypes = [vp, vp, u, i]
    for n in ("orc_execute_filter_input", "orc_delete_filter_input", "orc_delete_filter_output"):
        getattr(L, n).argtypes = [vp]
    L.orc_execute_filter_output.argtypes = [vp, i]
    L.orc_i0f.restype = f
    L.orc_i0f.argtypes = [f]
    L.orc_i1f.restype = f
    L.orc_i1f.argtypes = [f]
    L.orc_make_kaiser.argtypes = [vp, u, f]
    L.orc_window_filter.argtypes = [i, i, vp, f]
    L.orc_window_rfilter.argtypes = [i, i, vp, f]
    L.orc_noise_gain.restype = f
    L.orc_noise_gain.argtypes = [vp]
    L.orc_set_filter.argtypes = [vp, f, f, f]
    for n in ("orc_filter_in_input", "orc_filter_in_input_r", "orc_filter_in_fdomain", "orc_filter_out_output",
              "orc_filter_out_output_r", "orc_filter_out_response"):
        getattr(L, n).restype = vp
        getattr(L, n).argtypes = [vp]
    L.orc_filter_out_noise_gain.restype = f
    L.orc_filter_out_noise_gain.argtypes = [vp]
    L.orc_filter_out_set_type.argtypes = [vp, i]
    L.orc_filter_out_set_type.restype = None
    L.orc_notch_init.argtypes = [vp, d, f]
    L.orc_notch_block.argtypes = [vp, vp, i]
    L.orc_is_phasor_init.argtypes = [Cd]
    L.orc_set_osc.argtypes = [vp, d, d]
    L.orc_set_osc.restype = None
    L.orc_step_osc.restype = Cd
    L.orc_step_osc.argtypes = [vp]
    L.orc_renorm_osc.argtypes = [vp]
    L.orc_step_osc_block.argtypes = [vp, vp, i]
    L.orc_step_osc_block.restype = None
    L.orc_init_pll.argtypes = [vp, f, f, d, f]
    L.orc_init_pll.restype = None
    L.orc_run_pll.restype = f
    L.orc_run_pll.argtypes = [vp, f]
    L.orc_hb15_block.argtypes = [vp, vp, vp, i]
    L.orc_hb15_block.restype = None
    L.orc_hb3_block.argtypes = [vp, vp, vp, i]
    L.orc_hb3_block.restype = None
    L.orc_decimate_rail.argtypes = [vp, vp, i, i, vp, i]
    L.orc_decimate_rail.restype = None
    L.orc_hb15_init_f8.argtypes = [vp]
    L.orc_sampcount.argtypes = [i, i]
    L.orc_convert_samples.argtypes = [i, vp, i, f, vp]
    L.orc_mix_samples.restype = f
    L.orc_mix_samples.argtypes = [vp, vp, vp, vp, i]
    L.orc_skip_samples.argtypes = [vp, vp, i]
    L.orc_compute_n0.restype = f
    L.orc_compute_n0.argtypes = [vp, i, f, f, i]
    L.orc_fm_snr.restype = d
    L.orc_fm_snr.argtypes = [d]
    L.orc_demod_fm_block.argtypes = [vp, vp, vp, i, vp, vp]
    L.orc_demod_fm_block.restype = None
    L.orc_linear_init.argtypes = [vp, vp, f, d]
    L.orc_linear_init.restype = None
    L.orc_demod_linear_block.argtypes = [vp, vp, vp, i, vp, vp]
    L.orc_demod_linear_block.restype = None
    L.orc_am_init.argtypes = [vp, vp, f]
    L.orc_am_init.restype = None
    L.orc_demod_am_block.argtypes = [vp, vp, vp, i, vp, vp]
    L.orc_demod_am_block.restype = None
    L.orc_scaleclip.restype = C.c_short
    L.orc_scaleclip.argtypes = [f]
    L.orc_scaleclip_block.argtypes = [vp, vp, i]
    L.orc_packetise.argtypes = [vp, vp, i, i, vp, vp, vp, vp, i]
    L.orc_nextfastfft.restype = C.c_uint32
    L.orc_nextfastfft.argtypes = [C.c_uint32]
    L.orc_pcm_to_real.argtypes = [vp, vp, i]
    L.orc_pcm_to_real.restype = None
    L.orc_crc_good.argtypes = [vp, i]
    L.orc_afsk_init.argtypes = [vp]
    L.orc_afsk_init.restype = None
    L.orc_afsk_block.argtypes = [vp, vp, i, i, vp, i, vp, i, vp]
    L.orc_afsk_block_ms.argtypes = [vp, vp, vp, i, i, vp, i, vp, i, vp]
    L.orc_fe_init.argtypes = [vp, f]
    L.orc_fe_init.restype = None
    L.orc_hackrf_rx_block.argtypes = [vp, vp, i, i, f, vp, vp]
    L.orc_hackrf_rx_block.restype = None
    L.orc_funcube_block.argtypes = [vp, vp, i, f]
    L.orc_funcube_block.restype = None
    L.orc_hackrf_pack.argtypes = [i, vp, vp, i, f, vp]
    L.orc_hackrf_pack.restype = f
    L.orc_pl_init.argtypes = [vp]
    L.orc_pl_init.restype = None
    L.orc_pl_samples.argtypes = [vp, vp, i, vp, i, vp, vp, i, vp]
    L.orc_pl_samples.restype = None
    _libs[path] = L
    return L


def ref(portable=False):
    """The real reference objects (osc.c, decimate.c) compiled from /root/reference by oracle/Makefile."""
    path = REF_PORTABLE_SO if portable else REF_SO
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
    if not portable:
        L.set_osc.argtypes = [vp, d, d]
        L.set_osc.restype = None
        L.step_osc.restype = Cd
        L.step_osc.argtypes = [vp]
        L.renorm_osc.argtypes = [vp]
        L.is_phasor_init.argtypes = [Cd]
        L.init_pll.argtypes = [vp, f, f, d, f]
        L.init_pll.restype = None
        L.run_pll.restype = f
        L.run_pll.argtypes = [vp, f]
    L.hb15_block.argtypes = [vp, vp, vp, i]
    L.hb15_block.restype = None
    L.hb3_block.argtypes = [vp, vp, vp, i]
    L.hb3_block.restype = None
    if not portable and hasattr(L, "crc_good"):
        L.crc_good.argtypes = [vp, i]
    _libs[path] = L
    return L


# ---------------------------------------------------------------- numpy conveniences
def fft(x, sign=-1, fast=False):
    x = np.ascontiguousarray(x, np.complex64)
    y = np.empty_like(x)
    lib(fast).orc_fft(x.size, _vp(x), _vp(y), sign)
    return y


class Filter:
    """orc_filter_in + orc_filter_out driven like radio.c/fm.c drive the reference's."""

    def __init__(self, L, M, decimate, in_type=COMPLEX, out_type=COMPLEX, fast=False):
        self.lib = lib(fast)
        self.L, self.M, self.D = L, M, decimate
        self.N = L + M - 1
        self.ND = self.N // decimate
        self.in_type, self.out_type = in_type, out_type
        self.fin = self.lib.orc_create_filter_input(L, M, in_type)
        self.fout = self.lib.orc_create_filter_output(self.fin, None, decimate, out_type)
        self.olen = L // decimate

    def _view(self, ptr, count, dtype):
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count)

    def input(self):
        if self.in_type == REAL:
            return self._view(self.lib.orc_filter_in_input_r(self.fin), self.L, np.float32)
        return self._view(self.lib.orc_filter_in_input(self.fin), self.L, np.complex64)

    def fdomain(self):
        n = self.N // 2 + 1 if self.in_type == REAL else self.N
        return self._view(self.lib.orc_filter_in_fdomain(self.fin), n, np.complex64)

    def output(self):
        if self.out_type == REAL:
            return self._view(self.lib.orc_filter_out_output_r(self.fout), self.olen, np.float32)
        return self._view(self.lib.orc_filter_out_output(self.fout), self.olen, np.complex64)

    def response(self):
        n = self.ND // 2 + 1 if (self.in_type == REAL and self.out_type == REAL) else self.ND
        return self._view(self.lib.orc_filter_out_response(self.fout), n, np.complex64)

    def set_filter(self, low, high, beta):
        assert self.lib.orc_set_filter(self.fout, low, high, beta) == 0

    def noise_gain(self):
        return self.lib.orc_filter_out_noise_gain(self.fout)

    def set_out_type(self, t):
        self.out_type = t
        self.lib.orc_filter_out_set_type(self.fout, t)

    def execute_input(self, block):
        self.input()[:] = block
        self.lib.orc_execute_filter_input(self.fin)

    def execute_output(self, rotate=0):
        self.lib.orc_execute_filter_output(self.fout, rotate)
        return self.output().copy()

    def execute(self, block, rotate=0):
        self.execute_input(block)
        return self.execute_output(rotate)

    def close(self):
        if self.fout:
            self.lib.orc_delete_filter_output(self.fout)
            self.fout = None
        if self.fin:
            self.lib.orc_delete_filter_input(self.fin)
            self.fin = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sig_dict(s):
    return {k: getattr(s, k) for k, _ in Sig._fields_}


class FmDemod:
    def __init__(self, input_samprate, decimate, headroom, low, high, flat, fast=False):
        self.lib = lib(fast)
        self.p = FmParams(input_samprate, decimate, headroom, low, high, flat)
        self.st = FmState()
        self.sig = Sig()

    def block(self, y):
        y = np.ascontiguousarray(y, np.complex64)
        out = np.empty(y.size, np.float32)
        self.lib.orc_demod_fm_block(C.addressof(self.p), C.addressof(self.st), _vp(y), y.size, _vp(out),
                                    C.addressof(self.sig))
        return out, sig_dict(self.sig)


class LinearDemod:
    def __init__(self, output_samprate, pll=0, square=0, agc=0, env=0, channels=1, headroom=0.1778, hangtime=0.0,
                 recovery_rate=1.0, loop_bw=5.0, agc_gain=1.0, shift=0.0, fast=False):
        self.lib = lib(fast)
        self.p = LinearParams(output_samprate, pll, square, agc, env, channels, headroom, hangtime, recovery_rate,
                              loop_bw)
        self.st = LinearState()
        self.sig = Sig()
        self.lib.orc_linear_init(C.addressof(self.p), C.addressof(self.st), agc_gain, shift)

    def block(self, y):
        """returns (audio, sig); audio has olen floats (mono) or 2*olen interleaved I,Q (stereo)."""
        y = np.array(y, np.complex64)
        out = np.zeros(y.size, np.float32)
        self.lib.orc_demod_linear_block(C.addressof(self.p), C.addressof(self.st), _vp(y), y.size, _vp(out),
                                        C.addressof(self.sig))
        if not self.p.env and self.p.channels == 2:
            out = y.view(np.float32).copy()
        return out, sig_dict(self.sig)


class AmDemod:
    def __init__(self, input_samprate, decimate, output_samprate, headroom, hangtime, recovery_rate, agc_gain=1.0,
                 fast=False):
        self.lib = lib(fast)
        self.p = AmParams(input_samprate, decimate, output_samprate, headroom, hangtime, recovery_rate)
        self.st = AmState()
        self.sig = Sig()
        self.lib.orc_am_init(C.addressof(self.p), C.addressof(self.st), agc_gain)

    def block(self, y):
        y = np.ascontiguousarray(y, np.complex64)
        out = np.empty(y.size, np.float32)
        self.lib.orc_demod_am_block(C.addressof(self.p), C.addressof(self.st), _vp(y), y.size, _vp(out),
                                    C.addressof(self.sig))
        return out, sig_dict(self.sig)


def scaleclip(x, fast=False):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size, np.int16)
    lib(fast).orc_scaleclip_block(_vp(x), _vp(out), x.size)
    return out


class RtpPacketiser:
    """send_mono_output / send_stereo_output of audio.c:32-135 without the socket: for every call (one block of PCM
    words, already scaleclip'ed) the list of chunks (timestamp, seq, marker, send, offset, nwords).  Integer logic only."""
    PCM_BUFSIZE = 480  # audio.c:19

    def __init__(self, channels=1, timestamp=0, seq=0, silent=0):
        self.channels = channels
        self.timestamp, self.seq, self.silent = timestamp, seq, silent  # demod->output.rtp.*, output.silent
        self.packets = self.bytes = self.samples = 0

    def block(self, pcm_words, base_offset=0):
        pcm_words = np.asarray(pcm_words)
        out = []
        size = pcm_words.size // self.channels  # `size` frames (audio.c:32,83)
        pos = 0
        while size > 0:
            chunk = min(self.PCM_BUFSIZE, self.channels * size)  # audio.c:46,97
            words = pcm_words[pos:pos + chunk]
            not_silent = bool(np.any(words != 0))  # audio.c:48-52,99-103 (htons keeps zero zero)
            ts = self.timestamp & 0xFFFFFFFF
            self.timestamp = (self.timestamp + chunk // self.channels) & 0xFFFFFFFF  # audio.c:55,106
            marker = seq = 0
            if not_silent:  # audio.c:56-76,107-129
                self.packets += 1
                self.bytes += 2 * chunk
                marker = 1 if self.silent else 0
                self.silent = 0
                seq = self.seq
                self.seq = (self.seq + 1) & 0xFFFF
                self.samples += chunk // self.channels
            else:
                self.silent = 1  # audio.c:77-78,130-131
            out.append((ts, seq, marker, int(not_silent), base_offset + pos, chunk))
            pos += chunk
            size -= chunk // self.channels
        return out


def mix(doppler, lo2, x, fast=False):
    """radio.c:179-191 on a block; doppler may be None.  Returns (mixed, energy)."""
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty_like(x)
    e = lib(fast).orc_mix_samples(C.addressof(doppler) if doppler is not None else None, C.addressof(lo2), _vp(x),
                                  _vp(out), x.size)
    return out, e


def new_osc(f, r=0.0, fast=False):
    o = Osc()
    lib(fast).orc_set_osc(C.addressof(o), f, r)
    return o


# ---------------------------------------------------------------- packet.c: AFSK-1200 + HDLC per PCM session
AFSK_L, AFSK_M, AFSK_RATE = 1000, 1049, 48000.0  # packet.c:46-49


def crc_good(frame, fast=False):
    b = np.frombuffer(bytes(frame), np.uint8).copy()
    return bool(lib(fast).orc_crc_good(_vp(b), b.size))


class AfskSession:
    """One `struct session` of packet.c: big-endian PCM -> REAL filter input (packet.c:325,340-348) -> COMPLEX output,
    decimate 1, 100 Hz .. 4 kHz (407-408) -> decode_task's loop (437-545)."""

    def __init__(self, fast=False):
        self.L = lib(fast)
        self.fin = self.L.orc_create_filter_input(AFSK_L, AFSK_M, REAL)
        self.fout = self.L.orc_create_filter_output(self.fin, None, 1, COMPLEX)
        assert self.L.orc_set_filter(self.fout, 100.0 / AFSK_RATE, 4000.0 / AFSK_RATE, 3.0) == 0
        self.st = AfskState()
        self.L.orc_afsk_init(C.addressof(self.st))
        self.consumed = 0

    def block(self, pcm_be, maxframes=16, maxbits=64, y_override=None, ms_override=None):
        """pcm_be: AFSK_L network-order int16 words.  Returns (baseband[AFSK_L], frames, bits).
        y_override: run decode_task's loop over this baseband instead of the session's own filter output (the filter
        still runs and keeps its history): the sweeps' replay of the reference loop over the GPU's baseband."""
        pcm_be = np.ascontiguousarray(pcm_be, np.int16)
        assert pcm_be.size == AFSK_L
        p = self.L.orc_filter_in_input_r(self.fin)
        xin = np.frombuffer((C.c_char * (AFSK_L * 4)).from_address(p), np.float32, AFSK_L)
        self.L.orc_pcm_to_real(_vp(pcm_be), _vp(xin), AFSK_L)
        self.L.orc_execute_filter_input(self.fin)
        self.L.orc_execute_filter_output(self.fout, 0)
        q = self.L.orc_filter_out_output(self.fout)
        y = np.frombuffer((C.c_char * (AFSK_L * 8)).from_address(q), np.complex64, AFSK_L).copy()
        if y_override is not None:
            y = np.ascontiguousarray(y_override, np.complex64)
            assert y.size == AFSK_L
        frames = (Ax25Frame * maxframes)()
        bits = (AfskBit * maxbits)()
        nb = C.c_int(0)
        if ms_override is not None:  # the device's own tone products [AFSK_L, 4]: its arithmetic, operation for operation
            ms = np.ascontiguousarray(ms_override, np.float32)
            assert ms.size == 4 * AFSK_L
            nf = self.L.orc_afsk_block_ms(C.addressof(self.st), _vp(y), _vp(ms), AFSK_L, self.consumed, frames, maxframes, bits,
                                          maxbits, C.byref(nb))
        else:
            nf = self.L.orc_afsk_block(C.addressof(self.st), _vp(y), AFSK_L, self.consumed, frames, maxframes, bits, maxbits,
                                       C.byref(nb))
        self.consumed += AFSK_L
        fr = [dict(sample=f.sample, seq=f.seq, timestamp=f.timestamp, data=bytes(f.data[:f.bytes])) for f in frames[:nf]]
        bt = [(b.sample, b.cur_val) for b in bits[:min(nb.value, maxbits)]]
        return y, fr, bt

    def close(self):
        self.L.orc_delete_filter_output(self.fout)
        self.L.orc_delete_filter_input(self.fin)


# ---------------------------------------------------------------- pl.c: CTCSS and DTMF correlator banks per PCM session
PL_BLOCK, DTMF_BLOCK = 12000, 2400  # pl.c:26-27


def pl_tones(fast=False):
    return np.array(list((C.c_float * 55).in_dll(lib(fast), "orc_pl_tones")), np.float32)


class PlSessionRunner:
    def __init__(self, fast=False):
        self.L = lib(fast)
        self.sp = PlSession()
        self.L.orc_pl_init(C.addressof(self.sp))

    def samples(self, pcm_be):
        """Returns (pl results, dtmf results) of the integration intervals that end inside these samples."""
        pcm_be = np.ascontiguousarray(pcm_be, np.int16)
        maxpl, maxdt = pcm_be.size // PL_BLOCK + 2, pcm_be.size // DTMF_BLOCK + 2
        pl, dt = (PlResult * maxpl)(), (DtmfResult * maxdt)()
        npl, ndt = C.c_int(0), C.c_int(0)
        self.L.orc_pl_samples(C.addressof(self.sp), _vp(pcm_be), pcm_be.size, pl, maxpl, C.byref(npl), dt, maxdt, C.byref(ndt))
        return ([dict(tone=r.tone, snr=r.snr, e0=r.e0, e1=r.e1, index=r.index) for r in pl[:npl.value]],
                [dict(digit=chr(r.digit) if r.digit else "", low_index=r.low_index, high_index=r.high_index,
                      low_snr=r.low_snr, high_snr=r.high_snr) for r in dt[:ndt.value]])


# ---------------------------------------------------------------- front-end corrections (hackrf.c, funcube.c)
def fe_new(imbalance0=0.0, fast=False):
    st = FeState()
    lib(fast).orc_fe_init(C.addressof(st), imbalance0)
    return st


def fe_fields(st):
    return dict(dc_re=st.DC.re, dc_im=st.DC.im, sinphi=st.sinphi, imbalance=st.imbalance, gain_i=st.gain_i, gain_q=st.gain_q,
                secphi=st.secphi, tanphi=st.tanphi, rotate_phase=st.rotate_phase, clips=st.clips, in_power=st.in_power)


def hackrf_rx(st, iq, samples, nblocks, offset, rate_factor, fast=False):
    iq = np.ascontiguousarray(iq, np.int8)
    oi, oq = np.empty(samples * nblocks, np.float32), np.empty(samples * nblocks, np.float32)
    for b in range(nblocks):
        seg = np.ascontiguousarray(iq[2 * b * samples:2 * (b + 1) * samples])
        a, c = np.empty(samples, np.float32), np.empty(samples, np.float32)
        lib(fast).orc_hackrf_rx_block(C.addressof(st), _vp(seg), samples, offset, rate_factor, _vp(a), _vp(c))
        oi[b * samples:(b + 1) * samples], oq[b * samples:(b + 1) * samples] = a, c
    return oi, oq


def funcube_blocks(st, sampbuf, blocksize, nblocks, rate_factor, fast=False):
    buf = np.ascontiguousarray(sampbuf, np.int16).copy()
    for b in range(nblocks):
        seg = np.ascontiguousarray(buf[2 * b * blocksize:2 * (b + 1) * blocksize])
        lib(fast).orc_funcube_block(C.addressof(st), _vp(seg), blocksize, rate_factor)
        buf[2 * b * blocksize:2 * (b + 1) * blocksize] = seg
    return buf


def hackrf_pack(pt, ip, qp, atten, fast=False):
    ip, qp = np.ascontiguousarray(ip, np.float32), np.ascontiguousarray(qp, np.float32)
    out = np.zeros(ip.size * {97: 4, 95: 3, 98: 2}[pt], np.uint8)
    e = lib(fast).orc_hackrf_pack(pt, _vp(ip), _vp(qp), ip.size, atten, _vp(out))
    return out, e
