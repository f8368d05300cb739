#!/usr/bin/env python3
"""Soak of the cluster-resident bf16 LSTM kernels' hand-offs: the same launch N times, every output compared bit for bit
with the first — alone, with the members of a cluster spread over the XCDs, and with a competing stream of kernels on the
chip (uneven load: the condition under which visibility bugs show).  One line per (layer, mode)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
N = int(os.environ.get("REPS", 100))
NB = int(os.environ.get("NB", 64))


def layer(kind):
    b1 = kind == "full1"
    mode = "narrow" if kind == "narrow" else "full"
    H, c0, c2 = (256 if mode == "narrow" else 128), (16 if b1 else 256), (0 if b1 else 16)
    bidir = mode == "full"
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, bidir)], seed=3)
    w = [ops.pack_lstm_bf16w(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0, c2, dev)
         for s in ([""] + (["_reverse"] if bidir else []))]
    nt, nf = 300, (256 if mode == "narrow" else 257)
    x0 = torch.randn((NB, nt, nf, c0), device=dev) * 0.5
    x0 = x0 if b1 else x0.bfloat16()
    x2 = None if b1 else torch.randn((NB, nt, nf, c2), device=dev) * 0.5
    nd = 2 if bidir else 1

    def run():
        out = (torch.full((NB, nf, nt, H), float("nan"), device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3) if mode == "narrow"
               else torch.full((NB, nt, nf, nd * H), float("nan"), device=dev, dtype=torch.bfloat16))
        ops.lstm_layer(mode, x0, None, x2, w, H, out, bf16=True, wide=True)
        return out
    return run, (NB * (nf if mode == "narrow" else nt), H, nd)


side = torch.cuda.Stream()
a = torch.randn((4096, 4096), device=dev)
for kind in ("narrow", "full", "full1"):
    run, key = layer(kind)
    os.environ.pop("FNSSL_CLUSTER_SPREAD", None)
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    ref = run(); torch.cuda.synchronize()
    for label in ("alone", "spread over XCDs", "beside a competing stream"):
        if label == "spread over XCDs":
            os.environ["FNSSL_CLUSTER_SPREAD"] = "1"
            (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
        else:
            os.environ.pop("FNSSL_CLUSTER_SPREAD", None)
            (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
        bad = 0
        t0 = time.perf_counter()
        for i in range(N):
            if label == "beside a competing stream":
                with torch.cuda.stream(side):
                    for _ in range(1 + i % 3):                    # uneven: 1..3 matrix products of ~1 ms
                        a @ a
            out = run()
            torch.cuda.synchronize()
            bad += 0 if torch.equal(out, ref) else 1
        st = ops.lstm_cluster_status(*key, dev)
        print("%-7s %-26s %d runs: %d differ from the first, status word %d, %.2f ms per run (incl. sync + compare)"
              % (kind, label, N, bad, st, (time.perf_counter() - t0) / N * 1e3), flush=True)
