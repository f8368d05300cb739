#!/usr/bin/env python3
"""Per roofline kernel: HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_final_r04.sh
(pmc_c<cfg>_<COUNTER>.json, produced by tools/pmc_summary.py) -> the JSON bench.py reads (profiles/r04/hbm_traffic.json).
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide streaming reads at half their bytes
(MI355X_MICROARCH.md, HBM section): doubled here.  Counted at the fabric side of L2: Infinity-Cache hits are included."""
import json, os, sys
d = sys.argv[1]
KEYS = {2: ("c2_lstm_h256", "lstm_static3", 105696460800.0,
            "49152 sequences x 300 steps x 7 KiB per sequence-step: x_t and the residual operand read, h and h + skip written, h_{t-1} read back once, cell state both ways"),
        3: ("c3_lstm_h256", "lstm_bf16c_kernel<256", None, ""),
        4: ("c4_lstm_bwd_h256", "lstm_bwd2_kernel<256", None, ""),
        5: ("c5_sn_mamba_scan", "sn_mamba_scan", None, "")}
out = {}
for cfg, (key, pat, algo, note) in KEYS.items():
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(d, "pmc_c%d_%s.json" % (cfg, c))
        if not os.path.exists(p):
            continue
        try:
            j = json.load(open(p))
        except ValueError:
            continue
        v = n = 0
        names = []
        for k, e in j.items():
            if pat in k and c in e:
                v += e[c]
                n += e["launches"]
                names.append(k[:80])
        if n:
            tot[c] = (v / n, n, names)
    if len(tot) == 2:
        f, w = tot["FETCH_SIZE"][0], tot["WRITE_SIZE"][0]
        e = {"kernels": tot["FETCH_SIZE"][2], "launches": tot["FETCH_SIZE"][1], "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
             "bytes_per_launch": (2.0 * f + w) * 1024.0,
             "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --config %d --steps 1 --warmup 1`, per-launch "
                    "average over the kernel's launches, gfx950 2x read correction applied, counted at the fabric side of L2" % cfg}
        if algo:
            e["algorithmic_bytes_per_launch"] = algo
            e["algorithmic_bytes_are"] = note
            e["ratio_to_algorithmic"] = round(e["bytes_per_launch"] / algo, 3)
        out[key] = e
json.dump(out, sys.stdout, indent=1)
print()
