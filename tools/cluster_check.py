#!/usr/bin/env python3
"""Cluster-resident bf16 LSTM kernel (csrc/lstm_bf16c.h) against the pair-split kernels (FNSSL_NO_CLUSTER=1) on IPDnet's
narrow-band layer shape: bit-equality on small / ragged batches, repeatability, and timing at config 3's batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
LAYER = os.environ.get("LAYER", "narrow")            # narrow: H = 256 over time; full: H = 128, both directions, over frequency
B1 = LAYER == "full1"                                 # block 1's full-band layer: 16 fp32 feature channels only
if B1:
    LAYER = "full"
H, c0, c2 = (256 if LAYER == "narrow" else 128), (16 if B1 else 256), (0 if B1 else 16)
BIDIR = LAYER == "full"
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, BIDIR)], seed=1)
w = [ops.pack_lstm_bf16w(sd["L.weight_ih_l0" + sfx], sd["L.weight_hh_l0" + sfx], sd["L.bias_ih_l0" + sfx], sd["L.bias_hh_l0" + sfx], c0, c2, dev)
     for sfx in ([""] + (["_reverse"] if BIDIR else []))]
ND = 2 if BIDIR else 1


def run(x0, x2, cluster, reps=1):
    nb, nt, nf, _ = x0.shape
    if cluster:
        os.environ.pop("FNSSL_NO_CLUSTER", None)
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    else:
        os.environ["FNSSL_NO_CLUSTER"] = "1"
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    if LAYER == "narrow":
        out = torch.full((nb, nf, nt, H), float("nan"), device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    else:
        out = torch.full((nb, nt, nf, ND * H), float("nan"), device=dev, dtype=torch.bfloat16)
    for _ in range(reps):
        ops.lstm_layer(LAYER, x0, None, x2, w, H, out, bf16=True, wide=True)
    torch.cuda.synchronize()
    return out


mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "check"):
    for nb, nt, nf in ([(2, 6, 256), (5, 9, 256), (3, 40, 200), (16, 30, 256)] if LAYER == "narrow" else [(3, 256, 7), (5, 300, 9), (4, 333, 12), (16, 300, 20)]):
        g = torch.Generator(device="cpu").manual_seed(nb * 100 + nt)
        x0 = (torch.randn((nb, nt, nf, c0), generator=g) * 0.5).to(dev)
        x0 = x0 if B1 else x0.bfloat16()
        x2 = None if B1 else (torch.randn((nb, nt, nf, c2), generator=g) * 0.5).to(dev)
        a = run(x0, x2, True)
        b = run(x0, x2, False)
        a2 = run(x0, x2, True)
        d = (a.float() - b.float()).abs().max().item()
        print("nb %d nt %d nf %d: finite %s  max|cluster - pair| %.3g  equal %s  repeat-equal %s" %
              (nb, nt, nf, bool(torch.isfinite(a.float()).all()), d, torch.equal(a, b), torch.equal(a, a2)), flush=True)
if mode in ("all", "time"):
    for nb in [int(v) for v in os.environ.get("NBS", "64,32").split(",")]:
        nt, nf = 300, (256 if LAYER == "narrow" else 257)
        x0 = torch.randn((nb, nt, nf, c0), device=dev) * 0.5
        x0 = x0 if B1 else x0.bfloat16()
        x2 = None if B1 else torch.randn((nb, nt, nf, c2), device=dev) * 0.5
        for cluster in ((True, False) if not os.environ.get("AB") else (True, "ab", True, "ab")):
            if os.environ.get("AB"):
                if cluster == "ab":
                    os.environ[os.environ["AB"]] = "1"
                else:
                    os.environ.pop(os.environ["AB"], None)
                label = "B (%s=1)" % os.environ["AB"] if cluster == "ab" else "A"
                cluster = True
            else:
                label = "cluster" if cluster else "pair   "
            run(x0, x2, cluster, 2)
            t0 = time.perf_counter()
            run(x0, x2, cluster, 3)
            dt = (time.perf_counter() - t0) / 3
            fl = 2.0 * 4 * H * (c0 + c2 + H) * nb * nf * nt * ND
            print("nb %d %s: %.3f ms  %.0f TFLOP/s (%.2f of 2.5 PF)  %.2f us/step" %
                  (nb, label, dt * 1e3, fl / dt / 1e12, fl / dt / 2.5e15, dt / (nt if LAYER == "narrow" else nf) * 1e6), flush=True)
        a = run(x0, x2, True)
        b = run(x0, x2, False)
        print("nb %d equal %s" % (nb, torch.equal(a, b)), flush=True)
