#!/usr/bin/env python3
"""Soak of round 5's forms of the fp32 cluster kernel (csrc/lstm_f32c.h): H = 256 / clusters of 16 with the concatenated
segment (one 4-mic utterance), and the gate split of a one-group-per-cluster launch (2-mic utterance; a 12-frame chunk's
full-band layer with streaming-style uniform layout) — N launches each, every output compared bit for bit with the split
kernels' (FNSSL_NO_F32_SMALL), alone and beside a competing stream of matrix products; status word and fallback counter 0.

    REPS=300 python tools/f32c_small_soak.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch  # noqa: E402

from fnssl import _lib, ops  # noqa: E402
from fnssl import weights as W  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
N = int(os.environ.get("REPS", 200))
CASES = [  # name, mode, H, nb, nt, nf, c2, summed
    ("H=256 narrow, 4-mic utterance (6 groups per cluster of 16), [256 | 4] + residual", "narrow", 256, 6, 60, 256, 4, True),
    ("H=256 narrow, 96 pairs (96 groups per cluster: h streamed through the operand ring, 16 waves per member)", "narrow", 256, 96, 24, 256, 4, True),
    ("H=256 narrow, 2-mic utterance (1 group per cluster: gate split)", "narrow", 256, 1, 60, 256, 0, True),
    ("H=128 full, 2-mic utterance (1 group per cluster and direction: gate split)", "full", 128, 1, 249, 64, 0, True),
    ("H=128 full, 12-frame chunk of 6 pairs (72 evenly spaced sequences: gate split)", "full", 128, 6, 12, 64, 0, False),
]
side = torch.cuda.Stream()
a = torch.randn((4096, 4096), device=dev)
g = torch.Generator(device=dev)
g.manual_seed(11)
for name, mode, H, nb, nt, nf, c2, summed in CASES:
    bidir = mode == "full"
    ndir = 2 if bidir else 1
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(256 + c2, H, bidir)], seed=7 + H + c2)
    w = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], 256, c2, dev)
         for s in (("", "_reverse") if bidir else ("",))]
    x0 = torch.randn((nb, nt, nf, 256), generator=g, device=dev) * 0.5
    x2 = torch.randn((nb, nt, nf, c2), generator=g, device=dev) * 0.5 if c2 else None
    skip = torch.randn((nb, nt, nf, ndir * H), generator=g, device=dev) * 0.5 if summed else None

    def buf():
        if mode == "full":
            return torch.full((nb, nt, nf, ndir * H), float("nan"), device=dev)
        return torch.full((nb, nf, nt, ndir * H), float("nan"), device=dev).permute(0, 2, 1, 3)

    def run():
        out, osum = buf(), (buf() if summed else None)
        ops.lstm_layer(mode, x0, None, x2, w, H, out, skip=skip, out_sum=osum)
        return out, osum

    with _lib.tuning(no_f32_small=1):
        ref, refsum = run()
    torch.cuda.synchronize()
    fam = ops.lstm_plan(mode, x0, None, x2, w, H, buf(), skip=skip, out_sum=buf() if summed else None)
    nseq = nb * (nt if mode == "full" else nf)
    for label in ("alone", "beside a competing stream"):
        bad = 0
        ops.cluster_fallbacks(dev, reset=True)
        t0 = time.perf_counter()
        for i in range(N):
            if label != "alone":
                with torch.cuda.stream(side):
                    for _ in range(1 + i % 3):
                        a @ a
            out, osum = run()
            torch.cuda.synchronize()
            ok = torch.equal(out, ref) and (not summed or torch.equal(osum, refsum))
            bad += 0 if ok else 1
        print("%s [%s] %-26s %d launches: %d differ from the split kernels, status word %d, fallbacks %d, %.2f ms per launch incl. compare"
              % (name, fam[0], label, N, bad, ops.lstm_cluster_status(nseq, H, ndir, dev), ops.cluster_fallbacks(dev),
                 (time.perf_counter() - t0) / N * 1e3), flush=True)
