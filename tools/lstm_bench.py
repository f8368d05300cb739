#!/usr/bin/env python3
"""A/B the LSTM kernel variants on one layer at BASELINE config-2 size (GPU only).

    python tools/lstm_bench.py [--pairs 192] [--nt 300] [--nf 256] [--variants 1,2,3,4,5,6,7]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch  # noqa: E402

from fnssl import ops  # noqa: E402
from fnssl import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=192)
    ap.add_argument("--nt", type=int, default=300)
    ap.add_argument("--nf", type=int, default=256)
    ap.add_argument("--variants", default="0,4,5,8")
    ap.add_argument("--layers", default="narrow256s,full128s,narrow256_first,full128_first,narrow256,full128")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    nb, nt, nf = args.pairs, args.nt, args.nf
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    F = torch.randn((nb, nt, nf, 256), generator=g, device=dev) * 0.3
    N = torch.randn((nb, nf, nt, 256), generator=g, device=dev) * 0.3
    X = torch.randn((nb, nt, nf, 4), generator=g, device=dev)
    Nl = N.permute(0, 2, 1, 3)
    layers = {
        # name: (mode, x0, x1, x2, c0, c2, H, bidir, skip)
        "narrow256": ("narrow", F, Nl, None, 256, 0, 256, False, None),       # two-tensor input (FNblock API)
        "full128": ("full", Nl, F, None, 256, 0, 128, True, None),
        "narrow256s": ("narrow", F, None, None, 256, 0, 256, False, F),       # fused path: 1 input + residual out
        "full128s": ("full", Nl, None, None, 256, 0, 128, True, Nl),
        "narrow256_first": ("narrow", F, None, X, 256, 4, 256, False, F),
        "full128_first": ("full", X, None, None, 4, 0, 128, True, None),
    }
    for lname in args.layers.split(","):
        mode, x0, x1, x2, c0, c2, H, bidir, skip = layers[lname]
        sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, bidir)], seed=1)
        packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s],
                                sd["L.bias_hh_l0" + s], c0, c2, dev) for s in ([""] + (["_reverse"] if bidir else []))]
        ndir = 2 if bidir else 1
        out = torch.empty((nb, nf, nt, ndir * H), device=dev).permute(0, 2, 1, 3) if mode == "narrow" \
            else torch.empty((nb, nt, nf, ndir * H), device=dev)
        osum = torch.empty_like(out) if skip is not None else None
        if osum is not None and mode == "narrow":
            osum = torch.empty((nb, nf, nt, ndir * H), device=dev).permute(0, 2, 1, 3)
        nseq = nb * (nt if mode == "full" else nf)
        nsteps = nf if mode == "full" else nt
        flops = 2.0 * 4 * H * (c0 + c2 + H) * nseq * nsteps * ndir
        for v in [int(x) for x in args.variants.split(",")]:
            try:
                ops.lstm_layer(mode, x0, x1, x2, packed, H, out, v, skip=skip, out_sum=osum)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    ops.lstm_layer(mode, x0, x1, x2, packed, H, out, v, skip=skip, out_sum=osum)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                print("%-16s variant %d: %8.2f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)" %
                      (lname, v, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100), flush=True)
            except RuntimeError as ex:
                print("%-16s variant %d: FAILED %s" % (lname, v, ex), flush=True)


if __name__ == "__main__":
    main()
