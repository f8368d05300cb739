#!/usr/bin/env python3
"""Throughput of the IPDnet forward (BASELINE.json config 3 geometry, fp32) on one MI355X with a
per-kernel breakdown (HIP-event timing inside the library).  Secondary measurement — bench.py's
headline line stays the FN-SSL config-2 metric.

    python tools/ipdnet_bench.py [--nb 64] [--mics 8] [--hidden 256] [--frames 300] [--steps 3] [--offline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd", "IPDnet"))
import torch  # noqa: E402

import FixedAarryIPDnet as M  # noqa: E402
from fnssl import ops  # noqa: E402
from fnssl import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=64)
    ap.add_argument("--mics", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--offline", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="config 3 as written: bf16 weights / MFMA operands, fp32 accumulate")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    isz = 2 * args.mics
    sd = W.make_ipdnet_state(7, isz, args.hidden, 2, not args.offline)
    net = M.IPDnet(input_size=isz, hidden_size=args.hidden, max_track=2, is_online=not args.offline).eval()
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.to(dev)
    if args.bf16:
        net.bfloat16()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    ns = 256 * (args.frames + 1)
    sig = torch.randn((args.nb, ns, args.mics), generator=g, device=dev) * 0.1

    def step():
        return net(ops.preprocess_array(sig))

    for _ in range(args.warmup):
        y = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops.timing_enable(True)
    step()
    torch.cuda.synchronize()
    tm = ops.timing_collect()
    ops.timing_enable(False)
    kern = {k: {"ms": round(v["ms"], 3), "count": v["count"],
                "tflops": round(v["flops"] / v["ms"] / 1e9, 1) if v["flops"] > 0 and v["ms"] > 0 else None}
            for k, v in sorted(tm.items(), key=lambda kv: -kv[1]["ms"])}
    flops = sum(v["flops"] for v in tm.values())
    print(json.dumps({
        "metric": "TF-frames/sec IPDnet forward (fixed array), waveform -> DP-IPD",
        "value": round(args.nb * args.frames / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 2),
        "dtype": "bf16 (MFMA operands; fp32 accumulate and tensors)" if args.bf16 else "fp32", "data": "synthetic",
        "config": {"workload": "IPDnet %d-mic hidden %d %s, batch %d, 256 bins x %d frames"
                   % (args.mics, args.hidden, "offline" if args.offline else "online", args.nb, args.frames)},
        "tflops_per_step": round(flops / 1e12, 2), "achieved_tflops": round(flops / dt / 1e12, 1),
        ("frac_of_bf16_mfma_peak" if args.bf16 else "frac_of_fp32_mfma_peak"):
            round(flops / dt / (2500e12 if args.bf16 else 157.3e12), 3),
        "out_shape": list(y.shape), "finite": bool(torch.isfinite(y).all()), "kernels": kern}))


if __name__ == "__main__":
    main()
