#!/usr/bin/env python3
"""Soak of the fp32 cluster kernel's hand-offs (csrc/lstm_f32c.h) at config 2's full-band layer size: N launches, every
output compared bit for bit with the first; alone and beside a competing stream of matrix products."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
N = int(os.environ.get("REPS", 60))
H, c0 = 128, 256
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0, H, True)], seed=5)
w = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0, 0, dev) for s in ("", "_reverse")]
nb, nt, nf = 192, 300, int(os.environ.get("NF", 64))            # 57600 sequences x 2 directions, NF steps
x = torch.randn((nb, nt, nf, c0), device=dev) * 0.5
skip = torch.randn((nb, nt, nf, 2 * H), device=dev) * 0.5
out = torch.empty((nb, nt, nf, 2 * H), device=dev)
osum = torch.empty_like(out)


def run():
    out.fill_(float("nan")); osum.fill_(float("nan"))
    ops.lstm_layer("full", x, None, None, w, H, out, skip=skip, out_sum=osum)


run(); torch.cuda.synchronize()
ref, refsum = out.clone(), osum.clone()
side = torch.cuda.Stream()
a = torch.randn((4096, 4096), device=dev)
for label in ("alone", "beside a competing stream"):
    bad = 0
    t0 = time.perf_counter()
    for i in range(N):
        if label != "alone":
            with torch.cuda.stream(side):
                for _ in range(1 + i % 3):
                    a @ a
        run()
        torch.cuda.synchronize()
        bad += 0 if (torch.equal(out, ref) and torch.equal(osum, refsum)) else 1
    print("%-26s %d launches (%d sequences x 2 directions x %d steps): %d differ from the first, status word %d, %.1f ms per launch incl. fill + compare"
          % (label, N, nb * nt, nf, bad, ops.lstm_cluster_status(nb * nt, H, 2, dev), (time.perf_counter() - t0) / N * 1e3), flush=True)
