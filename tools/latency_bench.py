#!/usr/bin/env python3
"""Latency of the reference's real predict shapes (FN-SSL/Learner.py:219-272: one recording; online model = default):
one utterance (4-mic 'MM' / 2-mic, 300 / 249 frames) and a 12-frame streaming chunk, with the per-kernel breakdown and
the kernel family every LSTM layer takes (fnssl_lstm_plan).  Optional FNSSL_<KNOB>=... on the command line selects
A/B knobs (parsed by fnssl/_lib.py).

    python tools/latency_bench.py [--reps 20] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK = 157.3e12
FLOP_PER_TF_POINT = 4997120


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], ms[0]


def breakdown(fn, ops, reps=3):
    ops.timing_select(None)
    ops.timing_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ops.timing_enable(False)
    k = ops.timing_collect()
    return {n: round(v["ms"] / reps, 4) for n, v in sorted(k.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import predict_step as ps
    from fnssl import ops
    from fnssl import weights as W
    dev = torch.device("cuda:0")
    sd = W.make_fnssl_state(0, is_online=True)
    model = ps.MyModel(ch_mode="MM", device=str(dev))
    model.arch.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    net = model.arch
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    out = {"knobs": {k: v for k, v in os.environ.items() if k.startswith("FNSSL_")}, "cases": {}}
    for name, nb, nch, nt in (("utt_4mic_300", 1, 4, 300), ("utt_2mic_249", 1, 2, 249), ("utt_2mic_300", 1, 2, 300),
                              ("batch4_4mic_300", 4, 4, 300), ("batch16_4mic_300", 16, 4, 300)):
        ns = 512 + (nt - 1) * 256
        batch = torch.randn((nb, nch, ns), generator=g, device=dev)
        fn = lambda: model.predict_step(batch, 0)   # noqa: E731
        med, best = timed(fn, args.reps)
        npair = nch * (nch - 1) // 2
        flop = FLOP_PER_TF_POINT * 256.0 * npair * nb * nt
        out["cases"][name] = {"ms": round(med, 3), "ms_best": round(best, 3), "frames_per_s": round(nb * nt / med * 1e3, 1),
                              "frac_of_fp32_mfma_roof": round(flop / (med * 1e-3) / PEAK, 4), "kernels_ms": breakdown(fn, ops),
                              "fallbacks": ops.cluster_fallbacks(dev, reset=True)}
        print(name, json.dumps(out["cases"][name]), flush=True)
    # 12-frame streaming chunks (Model.FN_SSL.forward_stream), 4-mic 'MM' = 6 pairs and 2-mic = 1 pair
    for name, npair in (("chunk12_6pairs", 6), ("chunk12_1pair", 1)):
        x = torch.randn((npair, 4, 256, 12), generator=g, device=dev)
        state = {"s": None}

        def fn():
            _, state["s"] = net.forward_stream(x, state["s"])
        med, best = timed(fn, args.reps)
        out["cases"][name] = {"ms": round(med, 3), "ms_best": round(best, 3), "kernels_ms": breakdown(fn, ops),
                              "fallbacks": ops.cluster_fallbacks(dev, reset=True)}
        print(name, json.dumps(out["cases"][name]), flush=True)
    # which kernel family each layer of the one-utterance forward takes
    fams = {}
    for nm, nb_, nt_ in (("utt_4mic_300", 6, 300), ("chunk12_6pairs", 6, 12), ("utt_2mic_249", 1, 249)):
        x = torch.zeros((nb_, nt_, 256, 256), device=dev)
        x4 = torch.zeros((nb_, nt_, 256, 4), device=dev)
        b1, b2 = net.block_1._streams(dev), net.block_2._streams(dev)
        f = torch.empty((nb_, nt_, 256, 256), device=dev)
        n = torch.empty((nb_, 256, nt_, 256), device=dev).permute(0, 2, 1, 3)
        fams[nm] = {"full_b1": ops.lstm_plan("full", x4, None, None, b1[0], 128, f),
                    "narr_b1": ops.lstm_plan("narrow", x, None, x4, b1[1], 256, n),
                    "full_b23": ops.lstm_plan("full", x, None, None, b2[0], 128, f),
                    "narr_b23": ops.lstm_plan("narrow", x, None, None, b2[1], 256, n)}
    out["families"] = fams
    print("families", json.dumps(fams), flush=True)
    if args.json:
        with open(args.json, "w") as fjson:
            json.dump(out, fjson, indent=1)


if __name__ == "__main__":
    main()
