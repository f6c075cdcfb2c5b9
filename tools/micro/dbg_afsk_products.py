import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, orc, pyka9q as hip, synth, fuzz_afsk
assert hip.lib().ka9q_hip_init(0) == 0
rng = np.random.default_rng(3)
L = orc.AFSK_L
x, _ = fuzz_afsk.stream(rng, 0, 3)
dev = hip.Afsk(1, 3, max_frames=8, max_bits=3 * 26)
dev.write_pcm(0, x[:3 * L]); dev.execute(3)
gb = dev.bits(0); ms = dev.products(0, 3); gy = dev.baseband(0, 3)
o = orc.AfskSession()
bp = []
for b in range(3):
    _y, fr, bt = o.block(x[b * L:(b + 1) * L], ms_override=ms[b * L:(b + 1) * L]); bp += bt
print("gpu bits", [int(v) for v in gb["sample"]][:6], [float(v) for v in gb["cur_val"]][:6])
print("orc bits", [b[0] for b in bp][:6], [float(b[1]) for b in bp][:6])
# numpy replay of the first bit
f = np.float32
ma = np.zeros(2, f); sa = np.zeros(2, f)
for n in range(40):
    ma[0] = f(ma[0] + ms[n, 0]); ma[1] = f(ma[1] + ms[n, 1]); sa[0] = f(sa[0] + ms[n, 2]); sa[1] = f(sa[1] + ms[n, 3])
cur = f(f(f(ma[0] * ma[0]) + f(ma[1] * ma[1])) - f(f(sa[0] * sa[0]) + f(sa[1] * sa[1])))
print("numpy first bit", float(cur), "ms[0]", ms[0], "gy[0]", gy[0])
