#!/usr/bin/env python3
"""Throughput of the decimation cascade (decimate.c:113-163 as hackrf.c:495-509 calls it): 12.288 MS/s -> 192 kS/s, 64:1,
six 15-tap half-band stages on the I and Q rails.  BASELINE.md section 1 holds the reference's only published figure for
this row: 83 % of one Atom core at that rate.

   python3 tools/decimate_bench.py [log_decimate] [stage_threshold]

Prints one line per case: host-memory calls of one USB transfer's worth (the PCIe-inclusive path a drop-in caller gets)
and device-resident rails of growing length (the kernel alone), fused and staged."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ka9q-radio_amd"))
import numpy as np, torch, pyka9q as hip
LD = int(sys.argv[1]) if len(sys.argv) > 1 else 6
THR = int(sys.argv[2]) if len(sys.argv) > 2 else 8
FS_IN = 12.288e6
torch.cuda.set_device(0)
L = hip.lib()
assert L.ka9q_hip_init(0) == 0
rails = 2


def states():
    arr = (hip.Hb15State * (rails * LD))()
    for s in arr:
        for j, c in enumerate((-6.0, 33.0, -116.0, 490.0)):
            s.coeffs[j] = c / 802.0
    return arr


def run(tag, chunk, device, fused, reps):
    os.environ["KA9Q_HIP_DECIMATE_FUSED"] = "1" if fused else "0"
    n = chunk << LD
    sg, h3 = states(), np.zeros(rails * LD, np.float32)
    if device:
        x = torch.rand(rails, n, device="cuda") * 2 - 1
        ptr = x.data_ptr()
    else:
        xh = np.random.default_rng(1).uniform(-1, 1, (rails, n)).astype(np.float32)
        ptr = xh.ctypes.data
    for _ in range(3):
        assert L.kh_decimate_rails(sg, h3.ctypes.data, LD, THR, ptr, n, rails, chunk) == 0, hip.last_error()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        L.kh_decimate_rails(sg, h3.ctypes.data, LD, THR, ptr, n, rails, chunk)
    dt = (time.perf_counter() - t0) / reps
    alg = rails * 4 * (n + chunk)  # bytes: every raw float read once, every output written once
    print(f"DEC {tag}: 2^{LD}:1 thr {THR} chunk {chunk} ({n} raw samples/rail, {n / FS_IN * 1e3:.2f} ms of signal at 12.288 MS/s) "
          f"{'fused' if fused else 'staged'} {'device' if device else 'host'} rails: {dt * 1e6:.1f} us/call, "
          f"{n / dt / 1e6:.0f} MS/s complex in, realtime x{(n / FS_IN) / dt:.0f}, algorithmic {alg / dt / 1e9:.1f} GB/s", flush=True)


for fused in (True, False):
    run("usb-transfer", 2048, False, fused, 200)       # 131072 complex samples: one 262144-byte HackRF transfer (hackrf.c:458)
    run("resident-small", 2048, True, fused, 200)
    run("resident-64k", 65536, True, fused, 50)
    run("resident-1M", 1 << 20, True, fused, 10)
