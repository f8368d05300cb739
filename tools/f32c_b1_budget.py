#!/usr/bin/env python3
"""Round 6: what block 1's full-band layer (H = 128, 4 input channels: 132 MFMAs per group-step and slice) spends its time on —
`lstm_f32c_kernel<128, 0, 1, 0>` at config 2's size on the ABLATE build (`make ABLATE=1` -> libfnssl_hip_abl.so; results of
ablated runs are wrong by construction), one timing-ablation bit at a time and all of them together:

    FNSSL_LIB_PATH=fn-ssl_amd/csrc/libfnssl_hip_abl.so python tools/f32c_b1_budget.py

bits of FNSSL_F32C_ABL: 1 one group's addressing for all, 2 cheap gates, 4 no tag waits, 8 no input loads, 16 no recurrent-operand
loads, 32 no stores, 64 no cell-state loads, 128 no LDS record reads in the quads, 256 no tag loads / publishes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch  # noqa: E402
from fnssl import ops, weights as W  # noqa: E402
from fnssl import _lib  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
nb, nt, nf, H = 192, 300, 256, 128
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(4, H, True)], seed=1)
w = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], 4, 0, dev)
     for s in ("", "_reverse")]
g = torch.Generator(device=dev)
g.manual_seed(0)
x = torch.randn((nb, nt, nf, 4), generator=g, device=dev)
out = torch.empty((nb, nt, nf, 2 * H), device=dev)
flop = 2.0 * 4 * H * (4 + H) * nb * nt * nf * 2
fn = lambda: ops.lstm_layer("full", x, None, None, w, H, out)  # noqa: E731


def timed(reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


names = {0: "shipping arithmetic", 1: "one group's addressing for all", 2: "cheap gates (no transcendentals)", 4: "no tag waits",
         8: "no input loads", 16: "no recurrent-operand loads", 32: "no stores", 64: "no cell-state loads",
         128: "no LDS record reads in the quads", 256: "no tag loads / publishes", 4 | 16 | 256: "no hand-off at all (4 + 16 + 256)",
         2 | 4 | 8 | 16 | 32 | 64 | 256: "matrix stream + LDS reads + addressing only", 511: "everything off: the MFMAs alone"}
assert ops.lstm_plan("full", x, None, None, w, H, out)[0] == "f32_cluster"
base = None
print("lstm_f32c_kernel<128, 0, 1, 0> (block 1's full-band layer), %d sequences x %d steps x 2 directions; %.3f TFLOP per launch" %
      (nb * nt, nf, flop / 1e12))
for m in names:
    if m:
        os.environ["FNSSL_F32C_ABL"] = str(m)
    else:
        os.environ.pop("FNSSL_F32C_ABL", None)
    _lib.refresh_tuning()
    ms = timed()
    base = ms if base is None else base
    print("FNSSL_F32C_ABL=%-4d %-46s %7.2f ms  %6.1f TFLOP/s  %.3f of 157.3   %+6.2f ms" %
          (m, names[m], ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, ms - base), flush=True)
os.environ.pop("FNSSL_F32C_ABL", None)
