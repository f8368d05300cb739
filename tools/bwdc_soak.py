#!/usr/bin/env python3
"""Soak of the cluster-resident BPTT kernel: N launches at config 4's full-band layer size (ragged variant), every dA / dx
compared bit for bit with the first and with the split kernels; status word 0 every time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H = 128
for c0g, nb, nt, nf in ((256, 33, 301, 24), (0, 32, 300, 24)):
    c_in = c0g if c0g else 16
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c_in, H, True)], seed=11)
    sfx = ("", "_reverse")
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c_in, 0, dev) for s in sfx]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn((nb, nt, nf, c_in), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 2 * H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
    ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)

    def run():
        da = torch.full((nb, nt, nf, 8 * H), float("nan"), device=dev)
        dx = torch.full((nb, nt, nf, 2 * c0g), float("nan"), device=dev) if c0g else None
        _, _, word = ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, status=True)
        return da, dx, word
    os.environ["FNSSL_BWD_NO_CLUSTER"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    ref, refx, _ = run()
    del os.environ["FNSSL_BWD_NO_CLUSTER"]
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    assert ops.lstm_backward("full", reserve, dh, ref.clone(), None if refx is None else refx.clone(), bw, H, c0g, plan_only=True) == "bwd_cluster"
    bad = 0
    for i in range(N):
        a, xa, word = run()
        ok = word == 0 and torch.equal(a, ref) and (refx is None or torch.equal(xa, refx))
        bad += int(not ok)
    print("c0g=%d: %d launches, %d differ from the split kernels" % (c0g, N, bad), flush=True)
