#!/usr/bin/env python3
"""Per roofline kernel: HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_final_r06.sh
(pmc_<tag>_<COUNTER>.json, produced by tools/pmc_summary.py) -> the JSON bench.py reads (profiles/r06/hbm_traffic.json).
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide streaming reads at half their bytes
(MI355X_MICROARCH.md, HBM section): doubled here.  Counted at the fabric side of L2: Infinity-Cache hits are included.
Every entry carries the ALGORITHMIC bytes per launch it is to be read against (round-4 review: c3 / c4 / c5 had none)."""
import json
import os
import sys

d = sys.argv[1]
KIB = 1024.0
# tag -> (key, kernel-name pattern, algorithmic bytes per launch, what they are)
KEYS = {
    "c2": ("c2_lstm_h256", "lstm_static4", 49152 * 300 * 7 * KIB,
           "49152 sequences x 300 steps x 7 KiB per sequence-step: x_t and the residual operand read, h and h + skip written, "
           "h_{t-1} read back once, cell state both ways"),
    "c3": ("c3_lstm_h256", "lstm_bf16c_kernel<256", 16384 * 300 * 2112.0,
           "16384 sequences x 300 steps x 2112 B per sequence-step: x0 (256 fp32 channels) and x2 (16) read, h (256) written; the "
           "h records the members exchange are L2 traffic of the hand-off, not compulsory HBM bytes"),
    "c4": ("c4_lstm_bwd_h256", "lstm_bwd2_kernel<256", 8192 * 300 * 12288.0,
           "8192 sequences x 300 steps x 12 KiB per sequence-step: 6 KiB of forward reserve (i, f, g, o, c_t, c_{t-1}) and 1 KiB of "
           "dh read, 4 KiB of dA and 1 KiB of dx written"),
    "c5": ("c5_sn_mamba_scan", "sn_mamba_scan", (2 * 1024 * 250 + 14 * 1024 * 50) * 2456.0 / 16.0,
           "average over the 16 launches of a step: 1024 sequences x (250 frames in layer 0, 50 in layers 1-7) x 2456 B per "
           "sequence-frame: xz (2 x 192) and dbl_t (38) read, y (192) written"),
    "c2b1": ("c2b1_lstm_h256", "lstm_f32c_kernel<256", 1536 * 300 * 7 * KIB,
             "one 4-mic utterance: 1536 sequences x 300 steps x 7 KiB per sequence-step (the basis of c2_lstm_h256); every one of "
             "the 16 members of a cluster reads the whole x_t and h_{t-1} rows: L2 hits when the members keep together"),
    "c2M": ("c2M_lstm_h256", "lstm_f32c_kernel<256", 24576 * 300 * 7 * KIB,
            "the 'M' pairing of config 2's batch (96 pairs): 24576 sequences x 300 steps x 7 KiB per sequence-step (the basis of "
            "c2_lstm_h256); streamed-row form, 16 waves per member, 96 groups per cluster of 16 CUs"),
    "c2s": ("c2s_lstm_h128", "lstm_f32c_kernel<128", 72 * 256 * 2 * 3.5 * KIB,
            "a 12-frame chunk of 6 pairs: 72 sequences x 256 steps x 2 directions x 3.5 KiB per sequence-step (x_t 1 KiB read, h "
            "0.5 KiB written and read back, cell state both ways, residual operand / sum where fused)"),
}
out = {}
for tag, (key, pat, algo, note) in KEYS.items():
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(d, "pmc_%s_%s.json" % (tag, c))
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
                names.append(k[:90])
        if n:
            tot[c] = (v / n, n, names)
    if len(tot) == 2:
        f, w = tot["FETCH_SIZE"][0], tot["WRITE_SIZE"][0]
        e = {"kernels": tot["FETCH_SIZE"][2], "launches": tot["FETCH_SIZE"][1], "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
             "bytes_per_launch": (2.0 * f + w) * 1024.0,
             "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_final_r06.sh, tag %s), per-launch average over the "
                    "kernel's launches, gfx950 2x read correction applied, counted at the fabric side of L2" % tag,
             "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_are": note,
             "ratio_to_algorithmic": round((2.0 * f + w) * 1024.0 / algo, 3)}
        out[key] = e
json.dump(out, sys.stdout, indent=1)
print()
