#!/usr/bin/env python3
"""tools/resource_usage.py — registers / spills / LDS of every conv3d_f16_mfma instantiation, from `make -C surfacenet_amd/csrc asm`
(/tmp/sn_asm/resource_usage.txt: hipcc -Rpass-analysis=kernel-resource-usage). Prints the kernels that spill or use scratch, and any whose
template arguments contain one of the given substrings:   python tools/resource_usage.py ["3, 2, 4, 5" ...]"""
import re
import subprocess
import sys

t = open("/tmp/sn_asm/resource_usage.txt").read()
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
want = sys.argv[1:]
for b in blocks:
    name = b.split("\n")[0].split(" [")[0].strip()
    if "conv3d_f16_mfma" not in name:
        continue

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    m = re.search(r"conv3d_f16_mfma<([^>]*)>", dem)
    args = m.group(1) if m else dem[:80]
    row = dict(vgpr=g("VGPRs"), agpr=g("AGPRs"), spill=g("VGPRs Spill"), scratch=g(r"ScratchSize \[bytes/lane\]"), lds=g(r"LDS Size \[bytes/block\]"), occ=g(r"Occupancy \[waves/SIMD\]"))
    if row["spill"] > 0 or row["scratch"] > 0 or any(w in args for w in want):
        print("%-44s %s" % (args, row))
