#!/usr/bin/env python3
"""Training-step throughput (BASELINE.json config 4 geometry: 32 two-microphone utterances per GPU,
256 bins x 300 frames, fp32) with a per-kernel breakdown.  Single GPU, or one process per GPU under
torch.distributed.run (RCCL all-reduce of the flat gradient).  Secondary measurement — bench.py's headline
stays the inference metric.

    python tools/train_bench.py [--utts 32] [--mics 2] [--frames 300] [--steps 3] [--chunk-pairs N]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch  # noqa: E402

import Model  # noqa: E402
from fnssl import ops, train  # noqa: E402
from fnssl import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--mics", type=int, default=2)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk-pairs", type=int, default=0)
    ap.add_argument("--offline", action="store_true")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=dev)
    sd = W.make_fnssl_state(3, 4, 256, not args.offline)
    net = Model.FN_SSL(is_online=not args.offline)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.to(dev)
    eng = train.TrainEngine(net, seed=1, chunk_pairs=args.chunk_pairs or None)
    g = torch.Generator(device=dev)
    g.manual_seed(100 + rank)
    npair = args.mics * (args.mics - 1) // 2
    ns = 256 * (args.frames + 1)
    sig = torch.randn((args.utts, ns, args.mics), generator=g, device=dev) * 0.1
    gt = torch.tanh(torch.randn((args.utts, args.frames // 12, 512, npair), generator=g, device=dev))

    def step():
        x = ops.preprocess(sig, "MM", layout=1)          # [utts*np, 4, 256, nt]
        return eng.step(x, gt, sync_loss=False)

    losses = []
    for _ in range(args.warmup):
        losses.append(float(step().item()))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    losses.append(float(loss.item()))
    ops.timing_enable(True)
    step()
    torch.cuda.synchronize()
    tm = ops.timing_collect()
    ops.timing_enable(False)
    kern = {k: {"ms": round(v["ms"], 2), "count": v["count"],
                "tflops": round(v["flops"] / v["ms"] / 1e9, 1) if v["flops"] > 0 and v["ms"] > 0 else None}
            for k, v in sorted(tm.items(), key=lambda kv: -kv[1]["ms"])}
    # algorithmic flops of one step: forward + BPTT (same matmul volume) + weight-gradient GEMMs (again the same)
    fwd = 4997120.0 if not args.offline else 4210688.0
    tf_points = args.utts * npair * 256.0 * args.frames
    flops = 3.0 * fwd * tf_points
    if rank == 0:
        print(json.dumps({
            "metric": "training step (forward + backward + all-reduce + Adam), FN-SSL",
            "value": round(world * args.utts * args.frames / dt, 1), "unit": "utt-frames/s", "ms_per_step": round(dt * 1e3, 1),
            "n_gpus": world, "dtype": "fp32", "data": "synthetic", "scaling": "weak",
            "config": {"workload": "FN-SSL %s training step, %d utterances x %d mics per GPU, 256 bins x %d frames"
                       % ("offline" if args.offline else "online", args.utts, args.mics, args.frames),
                       "chunk_pairs": args.chunk_pairs},
            "tflops_per_step_per_gpu": round(flops / 1e12, 2), "achieved_tflops_per_gpu": round(flops / dt / 1e12, 1),
            "frac_of_fp32_mfma_peak": round(flops / dt / 157.3e12, 3),
            "losses": [round(v, 6) for v in losses],
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1), "kernels": kern}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
