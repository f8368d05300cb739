#!/usr/bin/env python3
"""Cluster-resident fp32 kernel of the H = 128 full-band layers (csrc/lstm_f32c.h) against the per-wave rounds of
lstm_static_kernel (FNSSL_NO_F32_CLUSTER=1): bit-equality on full-chip batches, timing at config 2's batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
H = 128


def make(c0, seed):
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0, H, True)], seed=seed)
    return [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0, 0, dev)
            for s in ("", "_reverse")]


def run(x, w, summed, cluster, reps=1):
    nb, nt, nf, _ = x.shape
    if cluster:
        os.environ.pop("FNSSL_NO_F32_CLUSTER", None)
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    else:
        os.environ["FNSSL_NO_F32_CLUSTER"] = "1"
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    out = torch.full((nb, nt, nf, 2 * H), float("nan"), device=dev)
    skip = (torch.arange(nb * nt * nf * 2 * H, device=dev, dtype=torch.float32).reshape(out.shape) % 7) * 0.125 if summed else None
    osum = torch.full_like(out, float("nan")) if summed else None
    for _ in range(reps):
        ops.lstm_layer("full", x, None, None, w, H, out, skip=skip, out_sum=osum)
    torch.cuda.synchronize()
    return out, osum


mode = sys.argv[1] if len(sys.argv) > 1 else "all"
for c0, summed in (((256, True),) if os.environ.get("ONLY256") else ((4, False),) if os.environ.get("ONLY4") else ((256, True), (4, False), (256, False))):
    w = make(c0, 11 + c0)
    if mode in ("all", "check"):
        for nb, nt, nf in ((96, 256, 6), (97, 300, 5)):               # 24576 / 29100 sequences: full-chip, the second ragged
            g = torch.Generator(device="cpu").manual_seed(nb + nf)
            x = (torch.randn((nb, nt, nf, c0), generator=g) * 0.5).to(dev)
            a, asum = run(x, w, summed, True)
            b, bsum = run(x, w, summed, False)
            a2, _ = run(x, w, summed, True)
            ok = torch.equal(a, b) and (not summed or torch.equal(asum, bsum))
            print("c0 %3d sum %d  nb %d nt %d nf %d: finite %s  max|cluster - rounds| %.3g  equal %s  repeat-equal %s"
                  % (c0, summed, nb, nt, nf, bool(torch.isfinite(a).all()), (a - b).abs().max().item(), ok, torch.equal(a, a2)), flush=True)
    if mode in ("all", "time") and not (c0 == 256 and not summed):
        nb, nt, nf = 192, 300, 256
        x = torch.randn((nb, nt, nf, c0), device=dev) * 0.5
        for cluster in (True, False, True, False):
            run(x, w, summed, cluster, 1)
            t0 = time.perf_counter()
            run(x, w, summed, cluster, 2)
            dt = (time.perf_counter() - t0) / 2
            fl = 2.0 * 4 * H * (c0 + H) * nb * nt * nf * 2
            print("c0 %3d  %s: %.2f ms  %.1f TFLOP/s (%.3f of 157.3)" % (c0, "cluster" if cluster else "rounds ", dt * 1e3, fl / dt / 1e12, fl / dt / 157.3e12), flush=True)
        del x
        torch.cuda.empty_cache()
