#!/usr/bin/env python3
"""Throughput of the IPDnet2 (OnlineSpatialNet) forward on one MI355X with a per-kernel breakdown (HIP-event timing
inside the library).  BASELINE.json config 5 mapping: 15 mics -> dim_input 30, 256 bins -> 2F = 512
outputs, online / causal path.  Secondary measurement; bench.py --config 5 prints the driver-format line.

    python tools/ipdnet2_bench.py [--nb 64] [--mics 15] [--frames 250] [--layers 8] [--steps 5]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from fnssl import ops  # noqa: E402
from fnssl import weights as W  # noqa: E402


def load_dropin():
    spec = importlib.util.spec_from_file_location("fnssl_ipdnet2_dropin", os.path.join(ROOT, "fn-ssl_amd", "IPDnet2", "IPDnet2.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def build(dev, mics, layers, seed=7):
    M = load_dropin()
    sd = W.make_ipdnet2_state(seed, dim_input=2 * mics, num_layers=layers)
    net = M.OnlineSpatialNet(dim_input=2 * mics, dim_output=16, num_layers=layers, dim_hidden=96, num_heads=4,
                             kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"],
                             dim_squeeze=8, num_freqs=256, attention="mamba(16,4)", rope=False, time_compression_layer=0,
                             fre_compression_ratio=16, time_compression_ratio=5).eval()
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return sd, net.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=64)
    ap.add_argument("--mics", type=int, default=15)
    ap.add_argument("--frames", type=int, default=250)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sd, net = build(dev, args.mics, args.layers)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    x = torch.randn((args.nb, 2 * args.mics, 256, args.frames), generator=g, device=dev)
    for _ in range(args.warmup):
        y = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops.timing_enable(True)
    net(x)
    torch.cuda.synchronize()
    tm = ops.timing_collect()
    ops.timing_enable(False)
    kern = {k: {"ms": round(v["ms"], 3), "count": v["count"],
                "tflops": round(v["flops"] / v["ms"] / 1e9, 1) if v["flops"] > 0 and v["ms"] > 0 else None}
            for k, v in sorted(tm.items(), key=lambda kv: -kv[1]["ms"])}
    from oracle import ipdnet2_oracle as O2
    fl = O2.flops_per_frame(dim_input=2 * args.mics, num_layers=args.layers)
    print(json.dumps({
        "metric": "utt-frames/sec IPDnet2 (OnlineSpatialNet) forward, features -> DP-IPD",
        "value": round(args.nb * args.frames / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "IPDnet2 %d-mic (dim_input %d), %d layers, batch %d, 256 bins x %d frames, online/causal"
                   % (args.mics, 2 * args.mics, args.layers, args.nb, args.frames)},
        "mflop_per_frame": round(fl / 1e6, 2), "achieved_tflops": round(fl * args.nb * args.frames / dt / 1e12, 2),
        "kernel_ms_sum": round(sum(v["ms"] for v in tm.values()), 3),
        "out_shape": list(y.shape), "finite": bool(torch.isfinite(y).all()), "kernels": kern}))


if __name__ == "__main__":
    main()
