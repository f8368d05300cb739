#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (…_counter_collection.csv) into per-kernel counter totals (JSON on stdout).

    python tools/pmc_summary.py gpurun_out/prof_x/p_counter_collection.csv [name-filter]
"""
import csv
import json
import re
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
        if flt and not re.search(flt, name):
            continue
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))[:110]
        acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[short].add(row.get("Dispatch_Id"))
out = {}
for k, v in acc.items():
    d = dict(v)
    d["launches"] = len(calls[k])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
        # MFMA busy cycles are summed over the 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE over the 8 XCDs
        d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES", 0) > 0:
        d["wait_any_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    out[k] = d
json.dump(out, sys.stdout, indent=1)
print()
