#!/usr/bin/env python3
"""Round 4: the cluster-resident BPTT kernel (lstm_bwdc.h) at config 4's full-band layer size against the split kernels,
and — on the ABLATE build (FNSSL_LIB_PATH=.../libfnssl_hip_abl.so) — with parts of it switched off (FNSSL_BWDC_ABLATE bits:
1 no tag waits, 2 no phase-A loads, 4 no phase-A stores, 8 no dA loads, 16 no output stores, 32 no drain before the
phase-B tag, 64 no phase A).  One launch per line: HIP-event time, fraction of the fp32 MFMA roof.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch
from fnssl import ops, weights as W

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
nb, nt, nf, H = 32, 300, 256, 128
ABL_LIB = "abl" in os.environ.get("FNSSL_LIB_PATH", "")


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device=dev)
g.manual_seed(0)
for c0g in (256, 0):
    c_in = c0g if c0g else 16
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c_in, H, True)], seed=3)
    sfx = ("", "_reverse")
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c_in, 0, dev) for s in sfx]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
    x = torch.randn((nb, nt, nf, c_in), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 2 * H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
    ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)
    da = torch.empty((nb, nt, nf, 8 * H), device=dev)
    dx = torch.empty((nb, nt, nf, 2 * c0g), device=dev) if c0g else None
    flops = 2.0 * 4 * H * (c0g + H) * nb * nt * nf * 2
    fn = lambda: ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g)  # noqa: E731

    def line(tag):
        ms = [timed(fn) for _ in range(2)]
        print("c0g=%-3d %-44s %s ms   %.3f of 157.3 TFLOP/s  [%s]" % (c0g, tag, " ".join("%.2f" % v for v in ms), flops / min(ms) / 1e9 / 157.3,
                                                                  ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, plan_only=True)), flush=True)

    os.environ["FNSSL_BWD_NO_CLUSTER"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    line("split kernels")
    del os.environ["FNSSL_BWD_NO_CLUSTER"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    os.environ["FNSSL_BWD_CLUSTER_MIN_GROUPS"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    line("cluster kernel")
    os.environ["FNSSL_BWD_CLUSTER_NO_ROTATE"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    line("cluster kernel, no rotation")
    del os.environ["FNSSL_BWD_CLUSTER_NO_ROTATE"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    os.environ["FNSSL_BWDC_NO_PREFETCH"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    line("cluster kernel, operands requested when needed")
    del os.environ["FNSSL_BWDC_NO_PREFETCH"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    os.environ["FNSSL_BWDC_NO_TOKEN"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    line("cluster kernel, no SIMD token")
    del os.environ["FNSSL_BWDC_NO_TOKEN"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    os.environ["FNSSL_BWDC_WAVES16"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    line("cluster kernel, 16 waves, 4-deep ring")
    del os.environ["FNSSL_BWDC_WAVES16"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    if ABL_LIB:
        for m in (128, 256, 384, 2, 4, 8, 1 | 8 | 16 | 32 | 64):
            os.environ["FNSSL_BWDC_ABLATE"] = str(m)
            (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
            line("cluster kernel, ablate %d" % m)
        del os.environ["FNSSL_BWDC_ABLATE"]
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    del os.environ["FNSSL_BWD_CLUSTER_MIN_GROUPS"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
