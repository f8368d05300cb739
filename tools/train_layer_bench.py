#!/usr/bin/env python3
"""Per-layer timing of the training LSTM kernels (reserve-saving forward, BPTT) at a given number of
microphone pairs; FNSSL_TRAIN_SPLIT=1|2|4 forces the wave-split geometry.

    python tools/train_layer_bench.py [--pairs 32] [--nt 300] [--nf 256]
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
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--nt", type=int, default=300)
    ap.add_argument("--nf", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    nb, nt, nf = args.pairs, args.nt, args.nf
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    layers = [("full  H128 c0=256", "full", 128, True, 256, 0, 256), ("full  H128 data4", "full", 128, True, 0, 4, 0),
              ("narrow H256 c0=256", "narrow", 256, False, 256, 0, 256), ("narrow H256 +4", "narrow", 256, False, 256, 4, 256)]
    for name, mode, H, bidir, c0, c2, c0g in layers:
        ndir = 2 if bidir else 1
        sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, bidir)], seed=1)
        sfx = [""] + (["_reverse"] if bidir else [])
        pk = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s],
                            sd["L.bias_hh_l0" + s], c0, c2, dev) for s in sfx]
        pkb = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev)
               for s in sfx]

        def nat(c):
            if mode == "full":
                return torch.zeros((nb, nt, nf, c), device=dev)
            return torch.zeros((nb, nf, nt, c), device=dev).permute(0, 2, 1, 3)
        x0 = nat(c0).normal_(generator=g) * 0.3 if c0 else None
        x2 = nat(c2).normal_(generator=g) if c2 else None
        out, da = nat(ndir * H), nat(ndir * 4 * H)
        dx = nat(ndir * c0g) if c0g else None
        dh = nat(ndir * H).normal_(generator=g)
        nseq, nsteps = (nb * nt, nf) if mode == "full" else (nb * nf, nt)
        res = torch.empty(ops.lstm_reserve_floats(nseq, H, ndir, nsteps), device=dev)
        for _ in range(2):
            ops.timing_enable(True)
            ops.lstm_layer(mode, x0, None, x2, pk, H, out, reserve=res)
            ops.lstm_backward(mode, res, dh, da, dx, pkb, H, c0g)
            torch.cuda.synchronize()
            tm = ops.timing_collect()
            ops.timing_enable(False)
        line = "%-20s split=%s" % (name, os.environ.get("FNSSL_TRAIN_SPLIT", "auto"))
        for k, v in sorted(tm.items()):
            line += "  %s %.2f ms %.1f TF" % (k, v["ms"], v["flops"] / v["ms"] / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
