#!/usr/bin/env python3
"""Round 4: per-phase cycle budget of lstm_bwdc_kernel (ABLATE build, FNSSL_BWDC_ABLATE=512): prints, for the six members of
cluster 0, waves 0 and 5, the shader cycles per group-step spent in each phase.  FNSSL_LIB_PATH=.../libfnssl_hip_abl.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0")
nb, nt, nf, H, c0g = 32, 300, 256, 128, 256
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0g, H, True)], seed=3)
sfx = ("", "_reverse")
packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0g, 0, dev) for s in sfx]
bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
g = torch.Generator(device=dev); g.manual_seed(0)
x = torch.randn((nb, nt, nf, c0g), generator=g, device=dev) * 0.7
dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
out = torch.empty((nb, nt, nf, 2 * H), device=dev)
reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)
da = torch.empty((nb, nt, nf, 8 * H), device=dev)
dx = torch.empty((nb, nt, nf, 2 * c0g), device=dev)
ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g)
torch.cuda.synchronize()
os.environ["FNSSL_BWDC_ABLATE"] = sys.argv[1] if len(sys.argv) > 1 else "512"
(lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g)
torch.cuda.synchronize()
