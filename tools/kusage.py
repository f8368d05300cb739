#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks read from stdin (one line per kernel)."""
import re
import sys

cur = None
rows = []
for line in sys.stdin:
    if "error" in line or "warning:" in line:
        print(line.rstrip()[:300])
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in [("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                     ("sgpr", r"TotalSGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")]:
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
filt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    name = r["name"]
    m = re.search(r"lstm_rec_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)", name)
    if m:
        name = "lstm H=%s NW=%s M=%s WMODE=%s HAS1=%s HAS2=%s" % m.groups()
    if filt and not re.search(filt, name):
        continue
    print("%-46s vgpr %3d agpr %3d sgpr %3d scratch %3d spill %3d occ %d lds %d" % (
        name[:46], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("spill", -1),
        r.get("occ", -1), r.get("lds", -1)))
