#!/usr/bin/env python3
"""Round 6: soak of lstm_static4_kernel's LDS-DMA weight ring (counted s_waitcnt vmcnt(K) in front of every ring barrier) at
config 2's narrow-band layer size, the three layer variants: N launches each, every output compared bit for bit with the
two-slice kernel's (FNSSL_NO_STATIC4=1, register-staged ring) — alone and beside a competing stream of matrix products (a
different arrival pattern of the DMA pieces).  A miscounted wait would show as a stale weight record = different bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch
from fnssl import ops, weights as W
from fnssl import _lib
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
N = int(os.environ.get("REPS", 40))
H, c0, nb, nf = 256, 256, 192, 256
nt = int(os.environ.get("NT", 40))          # steps per launch (49152 sequences)
side = torch.cuda.Stream()
a = torch.randn((4096, 4096), device=dev)
for c2, summed in ((0, True), (0, False), (4, True)):
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, False)], seed=7 + c2)
    w = [ops.pack_lstm(sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], sd["L.bias_ih_l0"], sd["L.bias_hh_l0"], c0, c2, dev)]
    x0 = torch.randn((nb, nt, nf, c0), device=dev) * 0.5
    x2 = torch.randn((nb, nt, nf, c2), device=dev) * 0.5 if c2 else None
    skip = torch.randn((nb, nt, nf, H), device=dev) * 0.5 if summed else None
    out = torch.empty((nb, nf, nt, H), device=dev).permute(0, 2, 1, 3)
    osum = torch.empty((nb, nf, nt, H), device=dev).permute(0, 2, 1, 3) if summed else None

    def run():
        out.fill_(float("nan"))
        if summed:
            osum.fill_(float("nan"))
        ops.lstm_layer("narrow", x0, None, x2, w, H, out, skip=skip, out_sum=osum)

    os.environ["FNSSL_NO_STATIC4"] = "1"
    _lib.refresh_tuning()
    run(); torch.cuda.synchronize()
    ref, refsum = out.clone(), (osum.clone() if summed else None)
    os.environ.pop("FNSSL_NO_STATIC4")
    _lib.refresh_tuning()
    assert ops.lstm_layer("narrow", x0, None, x2, w, H, out, skip=skip, out_sum=osum, plan_only=True)[0] == "static3"
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
            bad += 0 if (torch.equal(out, ref) and (not summed or torch.equal(osum, refsum))) else 1
        print("c2=%d summed=%d %-26s %d launches (49152 sequences x %d steps): %d differ from the two-slice kernel's output, %.1f ms per launch incl. fill + compare"
              % (c2, summed, label, N, nt, bad, (time.perf_counter() - t0) / N * 1e3), flush=True)
    del x0, x2, skip, out, osum, ref, refsum
    torch.cuda.empty_cache()
