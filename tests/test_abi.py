"""The C ABI of libka9q_hip without a GPU: the library loads, exports every function include/ka9q_hip.h declares,
the header is valid C and C++, the struct layouts that callers poke directly (filter.h:55-82, decimate.h:4-9) are
what the ctypes binding assumes, and every compute entry point fails loudly -- not silently on a CPU path -- when
there is no device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import pyka9q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ka9q_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}()]*(?:\([^()]*\)[^;{}()]*)*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_functions()
    assert len(names) >= 40, names
    L = pyka9q.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(pyka9q.EXPORTS) == names, set(pyka9q.EXPORTS) ^ set(names)


def test_header_is_valid_c_and_cxx_and_layouts_match_ctypes(tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "ka9q_hip.h"
#define P(s, f) printf(#s "." #f " %zu\n", offsetof(struct s, f))
int main(void) {
  printf("filter_in %zu\n", sizeof(struct filter_in));
  P(filter_in, in_type); P(filter_in, ilen); P(filter_in, impulse_length); P(filter_in, fdomain);
  P(filter_in, input_buffer); P(filter_in, input); P(filter_in, fwd_plan); P(filter_in, blocknum);
  P(filter_in, filter_mutex); P(filter_in, filter_cond); P(filter_in, fd);
  printf("filter_out %zu\n", sizeof(struct filter_out));
  P(filter_out, master); P(filter_out, out_type); P(filter_out, response); P(filter_out, response_mutex);
  P(filter_out, f_fdomain); P(filter_out, noise_gain); P(filter_out, output_buffer); P(filter_out, output);
  P(filter_out, rev_plan); P(filter_out, decimate); P(filter_out, olen); P(filter_out, blocknum);
  printf("hb15_state %zu\n", sizeof(struct hb15_state));
  printf("kh_bank_config %zu\n", sizeof(struct kh_bank_config));
  printf("kh_channel_config %zu\n", sizeof(struct kh_channel_config));
  P(kh_channel_config, shift);
  printf("kh_status %zu\n", sizeof(struct kh_status));
  printf("kh_input_status %zu\n", sizeof(struct kh_input_status));
  printf("kh_rtp_chunk %zu\n", sizeof(struct kh_rtp_chunk));
  printf("kh_rtp_state %zu\n", sizeof(struct kh_rtp_state));
  printf("kh_ax25_frame %zu\n", sizeof(struct kh_ax25_frame));
  P(kh_ax25_frame, data);
  printf("kh_afsk_bit %zu\n", sizeof(struct kh_afsk_bit));
  printf("kh_afsk_status %zu\n", sizeof(struct kh_afsk_status));
  printf("kh_pl_result %zu\n", sizeof(struct kh_pl_result));
  printf("kh_dtmf_result %zu\n", sizeof(struct kh_dtmf_result));
  printf("kh_fe_state %zu\n", sizeof(struct kh_fe_state));
  printf("enum %d %d %d %d\n", NONE, COMPLEX, CROSS_CONJ, REAL);
  return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)],
                   check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I",
                    os.path.join(ROOT, "include"), str(prog)], check=True)
    out = dict(line.rsplit(" ", 1) for line in subprocess.run([str(exe)], check=True, capture_output=True,
                                                              text=True).stdout.strip().splitlines()
               if not line.startswith("enum"))
    assert int(out["filter_in"]) == C.sizeof(pyka9q.FilterIn)
    assert int(out["filter_out"]) == C.sizeof(pyka9q.FilterOut)
    for name, _t in pyka9q.FilterIn._fields_:
        assert int(out[f"filter_in.{name}"]) == getattr(pyka9q.FilterIn, name).offset, name
    for name, _t in pyka9q.FilterOut._fields_:
        assert int(out[f"filter_out.{name}"]) == getattr(pyka9q.FilterOut, name).offset, name
    assert int(out["hb15_state"]) == C.sizeof(pyka9q.Hb15State) == 64
    assert int(out["kh_bank_config"]) == C.sizeof(pyka9q.BankConfig)
    assert int(out["kh_channel_config"]) == C.sizeof(pyka9q.ChannelConfig)
    assert int(out["kh_channel_config.shift"]) == pyka9q.ChannelConfig.shift.offset
    assert int(out["kh_status"]) == C.sizeof(pyka9q.Status) == pyka9q.STATUS_DTYPE.itemsize
    assert int(out["kh_input_status"]) == pyka9q.INPUT_STATUS_DTYPE.itemsize
    assert int(out["kh_rtp_chunk"]) == pyka9q.Bank.RTP_CHUNK.itemsize and int(out["kh_rtp_state"]) == pyka9q.Bank.RTP_STATE.itemsize
    assert int(out["kh_ax25_frame"]) == pyka9q.Afsk.FRAME.itemsize == 1040
    assert int(out["kh_ax25_frame.data"]) == pyka9q.Afsk.FRAME.fields["data"][1]
    assert int(out["kh_afsk_bit"]) == pyka9q.Afsk.BIT.itemsize and int(out["kh_afsk_status"]) == pyka9q.Afsk.STATUS.itemsize
    assert int(out["kh_pl_result"]) == pyka9q.Tones.PL.itemsize and int(out["kh_dtmf_result"]) == pyka9q.Tones.DTMF.itemsize
    assert int(out["kh_fe_state"]) == pyka9q.FE_STATE.itemsize


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present: the no-device behaviour cannot be observed")
def test_no_device_means_loud_failure_not_a_cpu_path():
    L = pyka9q.lib()
    assert L.ka9q_hip_init(0) == -1
    assert "no HIP device" in pyka9q.last_error() or "fallback" in pyka9q.last_error()
    assert not L.create_filter_input(3840, 4353, pyka9q.COMPLEX)
    assert not L.ka9q_hip_alloc(64)
    w = np.zeros(16, np.float32)
    assert L.make_kaiser(w.ctypes.data, 16, 3.0) == -1 and not w.any()
    r = np.ones(2048, np.complex64)
    assert L.window_filter(960, 1089, r.ctypes.data, 3.0) == -1
    cfg = pyka9q.BankConfig(3840, 4353, 4, 1, 1, 1, 192000, 0, 0, 0, 0.0, 0.0)
    assert not L.kh_bank_create(C.byref(cfg))
    with pytest.raises(RuntimeError):
        pyka9q.Bank(3840, 4353, 4, 1, 1, 1, 192000)
    # NULL-handle conventions of filter.c hold without a device too
    assert L.execute_filter_input(None) == -1 and L.execute_filter_output(None, 0) == -1
    assert L.delete_filter_input(None) == 0 and L.delete_filter_output(None) == 0
    assert L.kh_bank_destroy(None) == 0 and L.kh_bank_execute(None, 1) == -1
    # the packet / tone / front-end entry points fail the same way
    assert not L.kh_afsk_create(1, 1, 1, 0) and not L.kh_tones_create(1, 1)
    assert L.kh_afsk_destroy(None) == 0 and L.kh_tones_destroy(None) == 0
    st = pyka9q.fe_state(1.0)
    iq = np.zeros(8, np.int8)
    oi, oq = np.full(4, 7.0, np.float32), np.full(4, 7.0, np.float32)
    assert L.kh_hackrf_correct(st.ctypes.data, iq.ctypes.data, 4, 1, 1, 1e-7, oi.ctypes.data, oq.ctypes.data) == -1
    assert (oi == 7.0).all() and (oq == 7.0).all()


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under ka9q-radio_amd/ or include/ may name it."""
    bad = []
    for base in (os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "include")):
        for dirpath, _dirs, files in os.walk(base):
            if "build" in dirpath or "__pycache__" in dirpath:
                continue
            for fn in files:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                if re.search(r"ka9q_oracle|import orc\b|from orc\b|import pipeline|oracle/", text):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad
    deps = subprocess.run(["ldd", pyka9q.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps and "fftw" not in deps and "hipfft" not in deps and "rocfft" not in deps
