#!/usr/bin/env python3
"""Two forwards of one process on two HIP streams at the same time (a server handling two recordings): both run
cluster-resident kernels that want (nearly) every CU.  Could the dispatcher hand each launch part of the CUs, so that both
wait for members the other blocks until their bounded waits run out?  Measured (round 5): no — the launches take turns
by themselves: 54.2 ms per pair of one-utterance forwards = 2 x 27, no fallback.  (Chaining the cluster launches across
streams with events was built and removed: 77.5 ms per pair, worst 179 ms — an event hand-over per layer.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import predict_step as ps  # noqa: E402
from fnssl import _lib, ops  # noqa: E402
from fnssl import weights as W  # noqa: E402

dev = torch.device("cuda:0")
sd = W.make_fnssl_state(0, is_online=True)
models = []
for _ in range(2):
    m = ps.MyModel(ch_mode="MM", device=str(dev))
    m.arch.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    models.append(m.to(dev).eval())
g = torch.Generator(device=dev)
g.manual_seed(3)
batches = [torch.randn((1, 4, 512 + 299 * 256), generator=g, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
ref = [models[i].predict_step(batches[i], 0) for i in range(2)]
torch.cuda.synchronize()
N = int(os.environ.get("REPS", 10))
for label, knobs in (("default", {}), ("short bounded waits (~0.15 s)", {"cluster_spin_limit": 100000})):
    with _lib.tuning(**knobs):
        ops.cluster_fallbacks(dev, reset=True)
        outs = [None, None]
        worst = 0.0
        t_all = time.perf_counter()
        for _ in range(N):
            t0 = time.perf_counter()
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    outs[i] = models[i].predict_step(batches[i], 0)
            torch.cuda.synchronize()
            worst = max(worst, time.perf_counter() - t0)
            assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])
        dt = (time.perf_counter() - t_all) / N
        print("%-34s %d pairs of concurrent one-utterance forwards: %.1f ms per pair (worst %.1f ms), fallbacks %d"
              % (label, N, dt * 1e3, worst * 1e3, ops.cluster_fallbacks(dev)), flush=True)
