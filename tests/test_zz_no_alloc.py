"""kh_hackrf_pack_noalloc (no counterpart in the reference: for callers that cannot allocate device memory) against
kh_hackrf_pack, the existing entry point it is a variant of: payload bytes and the output energy must be identical
(same kernels, partial sums added in the same order).  Two calls in a row, the second with a smaller input: the static
device scratch keeps the first call's partial sums, and the second call must not see them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BPS = {97: 4, 95: 3, 98: 2}  # KH_PT_IQ16LE, KH_PT_IQ12, KH_PT_IQ8


class Dev:
    """device buffers through the HIP runtime the library itself is linked against (ctypes; no torch: its wheel brings
    a HIP runtime of its own, and two runtimes in one process do not share a device)"""
    rt = None

    def __init__(self, host):
        if Dev.rt is None:
            Dev.rt = C.CDLL("libamdhip64.so.7")
            Dev.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            Dev.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            Dev.rt.hipFree.argtypes = [C.c_void_p]
        self.host = np.ascontiguousarray(host)
        p = C.c_void_p()
        assert Dev.rt.hipMalloc(C.byref(p), max(self.host.nbytes, 16)) == 0
        self.ptr = p.value
        assert Dev.rt.hipMemcpy(self.ptr, self.host.ctypes.data, self.host.nbytes, 1) == 0  # hipMemcpyHostToDevice

    def get(self):
        out = np.empty_like(self.host)
        assert Dev.rt.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self):
        Dev.rt.hipFree(self.ptr)

    @staticmethod
    def sync():
        assert Dev.rt.hipDeviceSynchronize() == 0


def _pair(hip, payload, chunk, seed, atten):
    L = hip.lib()
    L.kh_hackrf_pack_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kh_hackrf_pack_noalloc.restype = C.c_int
    rng = np.random.default_rng(seed)
    # amplitudes that reach into the clip region of every format after x atten
    ip = Dev((rng.standard_normal(chunk) * 0.4).astype(np.float32))
    qp = Dev((rng.standard_normal(chunk) * 0.4).astype(np.float32))
    out_a = Dev(np.zeros(chunk * BPS[payload], np.uint8))
    out_b = Dev(np.full(chunk * BPS[payload] + 64, 0xA5, np.uint8))  # + guard bytes
    e_b = Dev(np.full(1, -1.0, np.float32))
    e_a = C.c_float()
    rc = L.kh_hackrf_pack(payload, ip.ptr, qp.ptr, chunk, atten, out_a.ptr, C.byref(e_a))
    assert rc == 0, hip.last_error()
    rc = L.kh_hackrf_pack_noalloc(payload, ip.ptr, qp.ptr, chunk, atten, out_b.ptr, e_b.ptr, None)
    assert rc == 0, hip.last_error()
    Dev.sync()  # the no-alloc call does not synchronise
    a, b = out_a.get(), out_b.get()
    assert np.array_equal(a, b[:a.size]), f"payload {payload} chunk {chunk}: bytes differ"
    assert (b[a.size:] == 0xA5).all(), "wrote past the payload"
    assert a.any()
    eb = float(e_b.get()[0])
    assert eb == e_a.value, (payload, chunk, eb, e_a.value)   # same partials, same order: bit-equal
    assert eb > 0
    for d in (ip, qp, out_a, out_b, e_b):
        d.free()


@pytest.mark.parametrize("payload", [97, 95, 98])
def test_noalloc_pack_matches_pack_twice_second_smaller(hip, payload):
    atten = {97: 32767.0, 95: 32767.0, 98: 127.0}[payload] * 1.5
    _pair(hip, payload, 300_007, 1, atten)   # 1172 workgroups, ragged tail
    _pair(hip, payload, 4_099, 2, atten)     # 17 workgroups: the scratch still holds 1155 stale partials
    _pair(hip, payload, 1 << 20, 3, atten)   # the maximum


def test_noalloc_pack_limits_and_errors(hip):
    L = hip.lib()
    L.kh_hackrf_pack_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kh_hackrf_pack_noalloc.restype = C.c_int
    n = 1024
    ip, qp = Dev(np.zeros(n, np.float32)), Dev(np.zeros(n, np.float32))
    out = Dev(np.zeros(4 * n, np.uint8))
    e = Dev(np.full(1, 7.0, np.float32))
    # beyond the maximum: refused before anything is launched
    assert L.kh_hackrf_pack_noalloc(97, ip.ptr, qp.ptr, (1 << 20) + 1, 1.0, out.ptr, e.ptr, None) == -1
    assert "maximum" in hip.last_error()
    # host pointers are refused (they would fault in the kernel)
    h = np.zeros(n, np.float32)
    assert L.kh_hackrf_pack_noalloc(97, h.ctypes.data, qp.ptr, n, 1.0, out.ptr, e.ptr, None) == -1
    assert "can address" in hip.last_error()
    assert L.kh_hackrf_pack_noalloc(11, ip.ptr, qp.ptr, n, 1.0, out.ptr, e.ptr, None) == -1
    # chunk 0: energy cleared, nothing else touched; energy pointer may be NULL
    assert L.kh_hackrf_pack_noalloc(97, ip.ptr, qp.ptr, 0, 1.0, out.ptr, e.ptr, None) == 0
    Dev.sync()
    assert float(e.get()[0]) == 0.0
    assert L.kh_hackrf_pack_noalloc(97, ip.ptr, qp.ptr, n, 1.0, out.ptr, None, None) == 0
    Dev.sync()
