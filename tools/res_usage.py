#!/usr/bin/env python3
"""usage: tools/res_usage.py <file.hip> [name filter]  -- VGPRs / spills / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage;
compiles for gfx950 on the CPU, no GPU needed)"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-x", "hip", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:70s} vgpr {v.get('VGPRs')} spill {v.get('VGPRs Spill')} scratch {v.get('ScratchSize [bytes/lane]')} occ {v.get('Occupancy [waves/SIMD]')} lds {v.get('LDS Size [bytes/block]')}")
