#!/usr/bin/env python3
"""Headline benchmark: DP-IPD forward throughput on BASELINE config 2
(4 mics, 257 bins x 300 frames, 32 utterances = 192 mic pairs per GPU, fp32).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path over one synthetic batch already
resident in HBM: waveforms -> STFT -> pair features -> 3 x (full-band BiLSTM,
narrow-band LSTM) -> DP-IPD head.  Every rank runs the same per-GPU batch (weak
scaling; utterances are independent, so there is no data-path collective).
Rank 0 prints ONE JSON line; `value` is utterance-frames per second over all
ranks.  The line also carries
  roofline     : fp32-MFMA roofline of the dominant kernel (narrow-band LSTM),
                 timed with HIP events on the launch stream inside the timed region;
  cpu_baseline : the PyTorch-CPU restatement of the reference (oracle/torch_ref.py)
                 timed on this host on a bounded sample (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 dense fp32
FLOP_PER_TF_POINT = {True: 4997120, False: 4210688}   # LSTM matmuls only, by is_online


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def usable_cores():
    """Threads the CPU baseline may use: affinity mask capped by the cgroup CPU quota (and by 64:
    oneDNN's LSTM does not scale past that on these shapes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nb", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--nch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--ch-mode", default="MM")
    ap.add_argument("--chunk-pairs", type=int, default=0)
    ap.add_argument("--offline", action="store_true", help="is_online=False (bidirectional narrow-band LSTM)")
    ap.add_argument("--bf16", action="store_true",
                    help="NOT the BASELINE metric: the optional fast mode (bf16 MFMA operands in the LSTMs, fp32 "
                         "accumulate/tensors); reported with dtype 'bf16' and its measured deviation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=1, help="utterances in the bounded CPU sample")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the bounded sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:       # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    import predict_step as ps
    from fnssl import ops
    from fnssl import weights as W

    online = not args.offline
    sd = W.make_fnssl_state(0, is_online=online)
    model = ps.MyModel(ch_mode=args.ch_mode, device=str(dev))
    if not online:
        import Model as at_model
        model.arch = at_model.FN_SSL(is_online=False)
    model.arch.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.arch.chunk_pairs = args.chunk_pairs
    model = model.to(dev).eval()
    if args.bf16:
        model.arch.bfloat16()          # optional fast mode, see --bf16

    ns = 512 + (args.frames - 1) * 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    batch = torch.randn((args.nb, args.nch, ns), generator=gen, device=dev, dtype=torch.float32)   # [nb, nch, ns]
    n_pairs = ops.num_pairs(args.nch, args.ch_mode)
    nt = ops.num_frames(ns)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    log("rank %d/%d on %s: %d utt x %d mics x %d frames, %d pairs" % (rank, world, torch.cuda.get_device_name(dev),
                                                                    args.nb, args.nch, nt, args.nb * n_pairs))
    out = None
    for _ in range(args.warmup):
        out = model.predict_step(batch, 0)
    sync_all()
    ops.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.predict_step(batch, 0)
    sync_all()
    dt = time.perf_counter() - t0
    ops.timing_enable(False)
    kern = ops.timing_collect()
    log("timed %d steps in %.3f s" % (args.steps, dt))
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert out is not None and tuple(out.shape) == (args.nb * n_pairs, nt // 12, 512)
    assert bool(torch.isfinite(out).all())

    utt_frames = args.nb * nt * args.steps * world
    value = utt_frames / dt
    flop_per_utt_frame = FLOP_PER_TF_POINT[online] * 256 * n_pairs

    # ---- roofline of the dominant kernel (narrow-band LSTM, H = 256) ------------------
    roof = None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01", "hbm_traffic_lstm_h256.json")
    if os.path.exists(tpath):       # PMC passes cannot run inside the timed region: measured by rocprofv3, committed
        with open(tpath) as f:
            traffic = json.load(f).get("bytes_per_launch")
    dom = kern.get("lstm_h256") if online else None
    if dom and dom["ms"] > 0 and args.bf16:
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "lstm_bf16_kernel<H=256> (narrow-band LSTM, bf16 MFMA operands)",
                "achieved": round(achieved, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
                "traffic": None, "launches": dom["count"], "avg_ms": round(dom["ms"] / max(1, dom["count"]), 3),
                "flop_per_launch": dom["flops"] / max(1, dom["count"])}
    elif dom and dom["ms"] > 0:
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "lstm_rec_kernel<H=256> (narrow-band LSTM)",
                "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                "launches": dom["count"], "avg_ms": round(dom["ms"] / max(1, dom["count"]), 3),
                "flop_per_launch": dom["flops"] / max(1, dom["count"])}
    breakdown = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": v["count"] / args.steps,
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flops"] else None}
                 for k, v in sorted(kern.items())}

    # ---- CPU baseline + same-run parity gate (rank 0, N = 1 only) ---------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_ref as R
        cores = usable_cores()
        torch.set_num_threads(cores)
        net = R.build(sd, online)
        sample = batch[:args.cpu_utts].cpu()
        R.predict_step(net, sample[:, :, :512 + 11 * 256], args.ch_mode)       # warm-up (12 frames)
        # probe on 24 frames, then size the timed sample to ~args.cpu_seconds of CPU work
        p0 = time.perf_counter()
        R.predict_step(net, sample[:, :, :512 + 23 * 256], args.ch_mode)
        per_frame = (time.perf_counter() - p0) / 24.0
        cpu_frames = int(min(nt, max(24, (args.cpu_seconds / max(per_frame, 1e-9)) // 12 * 12)))
        log("cpu baseline: %d threads, probe %.3f s/frame -> timing %d frames" % (cores, per_frame, cpu_frames))
        cns = 512 + (cpu_frames - 1) * 256
        c0 = time.perf_counter()
        ref_out = R.predict_step(net, sample[:, :, :cns], args.ch_mode)
        cdt = time.perf_counter() - c0
        cpu = {"value": round(args.cpu_utts * cpu_frames / cdt, 2), "unit": "frames/s", "cores": cores,
               "kind": "port",
               "sample": "%d utterance(s) x %d mics x %d frames ('%s', %d pairs), PyTorch CPU restatement of the "
                         "reference (torch.stft + oneDNN nn.LSTM), %d threads, %.1f s"
                         % (args.cpu_utts, args.nch, cpu_frames, args.ch_mode, args.cpu_utts * n_pairs, cores, cdt)}
        # same-run parity gate on the same waveforms (the narrow-band LSTM is causal and the
        # forgetting-norm is recursive, so the first cpu_frames frames do not depend on later ones
        # ... except through the full-band BiLSTM, which runs along frequency only: exact prefix)
        got = model.predict_step(batch[:args.cpu_utts, :, :cns], 0).cpu()
        err = (got - ref_out).abs()
        rt, at = (2e-2, 4e-3) if args.bf16 else (1e-4, 1e-5)
        parity = {"max_abs_err": float(err.max()), "rtol": rt, "atol": at, "frames": cpu_frames,
                  "ok": bool((err <= at + rt * ref_out.abs()).all())}
        log("parity vs CPU reference: max abs err %.3g ok=%s" % (parity["max_abs_err"], parity["ok"]))

    if rank == 0:
        line = {
            "metric": "TF-frames/sec DP-IPD forward, 4-mic 257-bin x 300-frame" +
                      (" [optional bf16 fast mode: NOT the BASELINE fp32 metric]" if args.bf16 else ""),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.bf16 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: FN-SSL (%s) DP-IPD forward," % ("online" if online else "offline") + " waveform->STFT->features->"
                                   "3x(full-band BiLSTM + narrow-band LSTM)->head; %d utterances/GPU x %d mics "
                                   "('%s' = %d pairs) x 257 bins x %d frames, fp32; frame = one STFT frame of one "
                                   "utterance" % (args.nb, args.nch, args.ch_mode, n_pairs, nt),
                       "utterances_per_gpu": args.nb, "mics": args.nch, "pairs_per_utterance": n_pairs,
                       "frames": nt, "bins": 257, "parallelism": "dp%d (utterance shards, no collective)" % world,
                       "chunk_pairs": args.chunk_pairs,
                       "gflop_per_frame": round(flop_per_utt_frame / 1e9, 3)},
            "whole_path_tflops": round(value * flop_per_utt_frame / 1e12 / world, 2),
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "kernels": breakdown,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
