#!/usr/bin/env python3
"""Time the config-3 narrow-band layer (H = 256 <- [256 bf16 | 16 fp32], 16384 sequences x 300 steps) through the wide
bf16 kernel; with a `make ABLATE=1` library and FNSSL_BF16W_ABL=<mask> the timing-ablation twins (wrong results)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
H, c0, c2 = 256, 256, 16
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, False)], seed=1)
w = [ops.pack_lstm_bf16w(sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], sd["L.bias_ih_l0"], sd["L.bias_hh_l0"], c0, c2, dev)]
nb, nt, nf = int(os.environ.get("NB", 64)), 300, 256
x0 = (torch.randn((nb, nt, nf, c0), device=dev) * 0.5).bfloat16()
x2 = torch.randn((nb, nt, nf, c2), device=dev) * 0.5
out = torch.empty((nb, nf, nt, H), device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
for _ in range(2): ops.lstm_layer("narrow", x0, None, x2, w, H, out, bf16=True, wide=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ops.lstm_layer("narrow", x0, None, x2, w, H, out, bf16=True, wide=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
fl = 2.0 * 4 * H * (c0 + c2 + H) * nb * nf * nt
print("ABL=%s  %.3f ms  %.0f TFLOP/s  (%.1f k cycles/step at 2.4 GHz)" % (os.environ.get("FNSSL_BF16W_ABL", "0"), dt * 1e3, fl / dt / 1e12, dt / nt * 2.4e6))
