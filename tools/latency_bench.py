#!/usr/bin/env python3
"""Latency of the small cases: one 4 s two-microphone utterance (BASELINE config 1 geometry) through
predict_step, and 12-frame streaming chunks through FN_SSL.forward_stream.  FNSSL_LSTM_SPLIT=1 disables the
several-waves-per-group geometry for comparison."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch  # noqa: E402

import predict_step as ps  # noqa: E402
from fnssl import ops  # noqa: E402
from fnssl import weights as W  # noqa: E402


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    m = ps.MyModel(device="cuda")
    m.arch.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in W.make_fnssl_state(0).items()})
    m.to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    out = {"split": os.environ.get("FNSSL_LSTM_SPLIT", "auto")}
    for nch in (2, 4):
        sig = torch.randn((1, nch, 64000), generator=g, device=dev) * 0.05
        out["utterance_4s_%dmic_ms" % nch] = round(timed(lambda: m.predict_step(sig, 0), 5), 2)
        x = ops.preprocess(sig.permute(0, 2, 1), "MM", layout=1)          # [np, 4, 256, 249]
        state = [None]

        def chunk():
            y, state[0] = m.arch.forward_stream(x[..., :12].contiguous(), state[0])
        out["stream_chunk_12frames_%dmic_ms" % nch] = round(timed(chunk, 20), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
