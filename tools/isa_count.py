#!/usr/bin/env python3
"""Static instruction census of the kernels in an AMDGPU .s file (hipcc -S --cuda-device-only).

usage: isa_count.py file.s [name-substring ...]
For straight-line kernels (the channel / finish wavefronts are almost fully unrolled) the static count per
category is the per-wavefront issue count, which is what bounds them.
"""
import re
import sys
from collections import Counter


def category(op):
    if op.startswith("v_"):
        if "_f64" in op:
            return "valu_f64"
        if op.startswith("v_pk_"):
            return "valu_pk"
        if re.match(r"v_(sqrt|rcp|rsq|sin|cos|exp|log)_", op):
            return "valu_trans"
        if "dpp" in op:
            return "valu_dpp"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    want = sys.argv[2:]
    kernels = {}
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur = m.group(1)
            kernels[cur] = Counter()
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section"):
            cur = None
            continue
        m = re.match(r"^([a-z][a-z0-9_]+)(\s|$)", s)
        if not m:
            if ";" in s and ("vgpr_count" in s or "ScratchSize" in s or "NumVgprs" in s or "Occupancy" in s or "LDSByteSize" in s):
                kernels[cur][s] = 0
            continue
        op = m.group(1)
        c = category(op)
        if " dpp" in s or "row_" in s or "quad_perm" in s:
            c = "valu_dpp" if c.startswith("valu") else c
        kernels[cur][c] += 1
        kernels[cur]["total"] += 1
    for name, cnt in kernels.items():
        if cnt["total"] < 20:
            continue
        if want and not any(w in name for w in want):
            continue
        print(name)
        cats = ["total", "valu", "valu_pk", "valu_f64", "valu_trans", "valu_dpp", "lds", "vmem", "smem", "salu", "waitcnt", "other"]
        print("  " + "  ".join(f"{c}={cnt[c]}" for c in cats if cnt[c]))
        for k in cnt:
            if k.startswith(";"):
                print("  " + k)


if __name__ == "__main__":
    main()
