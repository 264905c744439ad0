#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel (VGPRs, AGPRs, spills, occupancy, LDS).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt; kernel_regs.py res.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
blocks = re.split(r"remark: Function Name: ", txt)[1:]
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    short = re.sub(r"\(.*", "", name).replace("(anonymous namespace)::", "").replace("void ", "")
    if pats and not all(p in short for p in pats):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    vals = [g(k) for k in ("VGPRs", "AGPRs", "VGPRs Spill", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]")]
    print("%-70s VGPR %4s AGPR %4s spill %3s scratch %4s occ %s LDS %s" % ((short,) + tuple(vals)))
