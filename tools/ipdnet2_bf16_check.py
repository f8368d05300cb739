"""Calibration print-out for FNSSL_PRECISION_BF16 of the IPDnet2 row: deviations of the bf16 kernels from the oracle's
restatement of the rounding and from the fp32 oracle (run on the GPU box: python tools/ipdnet2_bf16_check.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fn-ssl_amd"), os.path.join(ROOT, "tests")]
import test_gpu_ipdnet2 as T  # noqa: E402
from conftest import rs_randn  # noqa: E402
from fnssl import spatialnet as sn  # noqa: E402
from oracle import ipdnet2_oracle as O2  # noqa: E402

dev = torch.device("cuda:0")


def err(name, got, want):
    e = np.abs(got - want)
    print("%-46s max %.3e  rms %.3e  (ref rms %.3e)" % (name, e.max(), np.sqrt((e ** 2).mean()), np.sqrt((want ** 2).mean())),
          flush=True)


sd, net = T.build_net(dev, 2500, dim_input=30, num_layers=2)
w0 = net.layers[0]._packed(dev)
for cin in (30, 10):
    w, b = rs_randn(1, (96, cin, 5), 0.1), rs_randn(2, (96,), 0.1)
    x = rs_randn(3, (2, cin, 16, 23))
    wT = T.to_dev(w, dev).permute(1, 2, 0).contiguous()
    got = sn.encoder(T.to_dev(x, dev), wT, T.to_dev(b, dev), precision=sn.BF16).cpu().numpy()
    with O2.bf16_products():
        want = np.stack([O2.causal_conv1d(x[:, :, f, :], w, b)[0] for f in range(16)], 1).transpose(0, 1, 3, 2)
    exact = np.stack([O2.causal_conv1d(x[:, :, f, :], w, b)[0] for f in range(16)], 1).transpose(0, 1, 3, 2)
    err("encoder cin %d: bf16 kernel vs bf16 oracle" % cin, got, want)
    err("encoder cin %d: bf16 kernel vs fp32 oracle" % cin, got, exact)
for nf, pool in ((256, 2), (128, 8), (16, 1)):
    x = rs_randn(10 + nf, (2, nf, 5, 96))
    got = sn.fconv(T.to_dev(x, dev), w0[0], residual=True, pool=pool, precision=sn.BF16).cpu().numpy()
    with O2.bf16_products():
        want = O2.avgpool_f(x + O2.fconv(sd, "layers.0.fconv1", x), pool)
    err("fconv nf %d pool %d: vs bf16 oracle" % (nf, pool), got, want)
    err("fconv nf %d pool %d: vs fp32 oracle" % (nf, pool), got, O2.avgpool_f(x + O2.fconv(sd, "layers.0.fconv1", x), pool))
x = rs_randn(2520, (3, 35, 96))
xs = T.to_dev(x, dev).unsqueeze(0)
with O2.bf16_products():
    want, _ = O2.mamba_block(sd, "layers.0.norm_mhsa", "layers.0.mhsa", x[None])
exact, _ = O2.mamba_block(sd, "layers.0.norm_mhsa", "layers.0.mhsa", x[None])
got = sn.mamba(xs, w0[3], residual=False, precision=sn.BF16).cpu().numpy()
err("mamba: vs bf16 oracle", got, want)
err("mamba: vs fp32 oracle", got, exact)
got5 = sn.mamba(xs, w0[3], residual=True, time_pool=5, precision=sn.BF16).cpu().numpy()
err("mamba + residual + pool 5: vs bf16 oracle", got5, O2.avgpool_t(x[None] + want, 5))
for layers, frames in ((3, 40), (8, 20)):
    sd, net = T.build_net(dev, 2600, dim_input=30, num_layers=layers)
    x = rs_randn(2601, (1, 30, 256, frames), 0.7)
    xd = T.to_dev(x, dev)
    ref32 = net(xd).cpu().numpy()
    out = net.bfloat16()(xd).cpu().numpy()
    sdb = {k: O2.bf16_round(v) for k, v in sd.items()}
    with O2.bf16_products():
        want = O2.forward(sdb, x)
    exact = O2.forward(sd, x)
    err("network %d layers: bf16 vs bf16 oracle" % layers, out, want)
    err("network %d layers: bf16 vs fp32 oracle" % layers, out, exact)
    err("network %d layers: bf16 vs fp32 kernels" % layers, out, ref32)
    err("network %d layers: fp32 kernels vs fp32 oracle" % layers, ref32, exact)
    e = np.abs(out - exact)
    for rt, at in ((2e-2, 4e-3), (2e-2, 1e-2), (2e-2, 1.5e-2)):
        print("   excess over rtol %g atol %g: %.3e" % (rt, at, (e - (at + rt * np.abs(exact))).max()))
