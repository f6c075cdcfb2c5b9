#!/usr/bin/env python3
"""front-end corrections with device-resident buffers: python3 tools/frontend_bench.py   (not the contract bench)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, torch, pyka9q
torch.cuda.set_device(0)
L = pyka9q.lib()
assert L.ka9q_hip_init(0) == 0
samples, nblocks = 131072, 64   # 64 hackrf transfers of 262144 bytes
g = torch.Generator(device="cuda"); g.manual_seed(1)
iq = torch.randint(-100, 100, (2 * samples * nblocks,), device="cuda", dtype=torch.int8, generator=g)
rails = torch.empty(2, samples * nblocks, device="cuda", dtype=torch.float32)  # Sampbuffer_i, Sampbuffer_q
oi, oq = rails[0], rails[1]
st = pyka9q.fe_state(1.0)
def step():
    rc = L.kh_hackrf_correct(st.ctypes.data, iq.data_ptr(), samples, nblocks, 1, 1e-7, oi.data_ptr(), oq.data_ptr())
    assert rc == 0, pyka9q.last_error()
for _ in range(3): step()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
dt = (time.perf_counter() - t0) / K
n = samples * nblocks
print(f"hackrf corrections: {nblocks} transfers x {samples} samples in {dt*1e6:.1f} us per call (synchronous ABI, device buffers): "
      f"{n/dt/1e9:.2f} GS/s, {n*(2*2+8)/dt/1e9:.0f} GB/s algorithmic (2 x 2 B read, 8 B written per sample); "
      f"state: imbalance {st['imbalance'][0]:.4f} clips {st['clips'][0]}")
out = torch.empty(4 * n, device="cuda", dtype=torch.uint8)
e = C.c_float()
def pack():
    assert L.kh_hackrf_pack(pyka9q.KH_PT_IQ16LE, oi.data_ptr(), oq.data_ptr(), n, 1000.0, out.data_ptr(), C.byref(e)) == 0
for _ in range(3): pack()
t0 = time.perf_counter()
for _ in range(K): pack()
dt = (time.perf_counter() - t0) / K
print(f"hackrf pack (IQ_PT): {n} samples in {dt*1e6:.1f} us per call: {n/dt/1e9:.2f} GS/s, {n*12/dt/1e9:.0f} GB/s algorithmic")

# the whole front end on device pointers: corrections -> six hb15 stages (64:1, hackrf.c:497-509) -> IQ_PT payload
LOGD = 6
hb = np.zeros((2, LOGD), dtype=[("coeffs", "<f4", 4), ("even", "<f4", 4), ("odd", "<f4", 4), ("old", "<f4", 4)])
hb["coeffs"] = np.array([-6, 33, -116, 490], np.float32) / np.float32(802)  # hackrf.c:442-451
hb3 = np.zeros((2, LOGD), np.float32)
L.kh_decimate_rails.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
chunk = n >> LOGD
payload = torch.empty(4 * chunk, device="cuda", dtype=torch.uint8)
def chain():
    step()
    assert L.kh_decimate_rails(hb.ctypes.data, hb3.ctypes.data, LOGD, 8, rails.data_ptr(), n, 2, chunk) == 0, pyka9q.last_error()
    assert L.kh_hackrf_pack(pyka9q.KH_PT_IQ16LE, rails[0].data_ptr(), rails[1].data_ptr(), chunk, 32767.0 * 0.5 ** LOGD,
                            payload.data_ptr(), C.byref(e)) == 0
for _ in range(3): chain()
t0 = time.perf_counter()
for _ in range(K): chain()
dt = (time.perf_counter() - t0) / K
print(f"correct -> 64:1 half-band cascade -> pack on device pointers: {n} input samples in {dt*1e6:.1f} us per pass: {n/dt/1e9:.2f} GS/s "
      f"(hackrf runs at 0.008 - 0.02 GS/s); output energy {e.value:.3e}, payload bytes nonzero {int((payload != 0).sum())}")
