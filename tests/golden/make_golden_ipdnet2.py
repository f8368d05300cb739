#!/usr/bin/env python3
"""Golden vectors for the IPDnet2 row (G14), generated from the REAL reference /root/reference/IPDnet2/IPDnet2.py
and arch/base/norm.py in the build container.  Data only: seeds, shapes and reference outputs.

``mamba_ssm`` is not installed (and pinned nowhere in the reference), and ``SpatialNetLayer.__init__`` calls
``Mamba(...)`` unconditionally (IPDnet2.py:127,132).  To run the reference's own layer / network code at all
this script registers a module named ``mamba_ssm`` whose ``Mamba`` is a torch transcription of the published
algorithm (same parameter names as the package).  Consequences, stated in oracle/ipdnet2_oracle.py too:

  * piece fixtures (LayerNorm, CausalConv1d, _fconv, _full, poolings, FreqInverse) come from the
    reference's own pure-torch code — PINNED;
  * layer / network fixtures pin the reference's ORCHESTRATION (residuals, permutes, poolings, output
    re-ordering) around the block; the block itself stays "parity unpinned".
"""
import contextlib
import io
import math
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
sys.path.insert(0, "/root/reference/IPDnet2")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402


class Mamba(nn.Module):
    """Published Mamba block (Gu & Dao 2023), parameter names of mamba_ssm.Mamba; sequential reference scan."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, layer_idx=None):
        super().__init__()
        self.d_inner, self.d_state, self.d_conv = expand * d_model, d_state, d_conv
        self.dt_rank = math.ceil(d_model / 16)
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=False)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1, bias=True)
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True)
        self.A_log = nn.Parameter(torch.zeros(self.d_inner, d_state))
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=False)

    def forward(self, hidden, inference_params=None):
        S, T, _ = hidden.shape
        xz = self.in_proj(hidden)
        x, z = xz[..., :self.d_inner], xz[..., self.d_inner:]
        u = Fn.silu(self.conv1d(x.transpose(1, 2))[..., :T]).transpose(1, 2)
        dbl = self.x_proj(u)
        dt, Bm, Cm = torch.split(dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = Fn.softplus(self.dt_proj(dt))
        A = -torch.exp(self.A_log)
        h = hidden.new_zeros(S, self.d_inner, self.d_state)
        ys = []
        for t in range(T):
            h = torch.exp(dt[:, t, :, None] * A) * h + dt[:, t, :, None] * Bm[:, t, None, :] * u[:, t, :, None]
            ys.append((h * Cm[:, t, None, :]).sum(-1) + self.D * u[:, t])
        y = torch.stack(ys, 1) * Fn.silu(z)
        return self.out_proj(y)


_m = types.ModuleType("mamba_ssm")
_m.Mamba = Mamba
_g = types.ModuleType("mamba_ssm.utils.generation")
_g.InferenceParams = type("InferenceParams", (), {"__init__": lambda self, *a, **k: None})
sys.modules["mamba_ssm"] = _m
sys.modules["mamba_ssm.utils"] = types.ModuleType("mamba_ssm.utils")
sys.modules["mamba_ssm.utils.generation"] = _g

import IPDnet2 as ref  # noqa: E402  (reference)
from arch.base.norm import LayerNorm as RefLayerNorm  # noqa: E402  (reference)
from fnssl import weights as W  # noqa: E402

torch.set_num_threads(8)


def rs_randn(seed, shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(np.float32)


def build(seed, **cfg):
    sd = W.make_ipdnet2_state(seed, **cfg)
    net = ref.OnlineSpatialNet(dim_input=cfg.get("dim_input", 10), dim_output=cfg.get("dim_output", 16),
                               num_layers=cfg.get("num_layers", 8), dim_hidden=cfg.get("dim_hidden", 96), num_heads=4,
                               kernel_size=(5, 3), conv_groups=(8, 8), norms=["LN", "LN", "GN", "LN", "LN", "LN"],
                               dim_squeeze=cfg.get("dim_squeeze", 8), num_freqs=cfg.get("num_freqs", 256),
                               attention="mamba(16,4)", rope=False, time_compression_layer=0,
                               fre_compression_ratio=16, time_compression_ratio=5).eval()
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})   # strict: names and shapes match
    return sd, net


@torch.no_grad()
def main():
    arrs = {}
    T = torch.from_numpy
    # ---- pieces, reference classes only -------------------------------------------------------------
    ln = RefLayerNorm(seq_last=True, normalized_shape=96).eval()
    ln.weight.copy_(T(rs_randn(1, (96,)) * 0.2 + 1)), ln.bias.copy_(T(rs_randn(2, (96,)) * 0.1))
    arrs["ln_w"], arrs["ln_b"] = ln.weight.numpy().copy(), ln.bias.numpy().copy()
    arrs["ln_out"] = ln(T(rs_randn(3, (3, 96, 7)))).numpy()                  # [B, H, Seq]
    cc = ref.CausalConv1d(in_channels=10, out_channels=96, kernel_size=5, look_ahead=0).eval()
    cc.weight.copy_(T(rs_randn(4, (96, 10, 5), 0.15))), cc.bias.copy_(T(rs_randn(5, (96,), 0.1)))
    xcc = T(rs_randn(6, (4, 10, 23)))
    arrs["cc_w"], arrs["cc_b"] = cc.weight.numpy().copy(), cc.bias.numpy().copy()
    arrs["cc_out"] = cc(xcc).numpy()
    # (the reference's carried-state branch, IPDnet2.py:72-74, cannot run: `-self.kernel_size` negates a tuple;
    #  the streaming tests therefore check chunked == whole-signal instead)

    sd, net = build(2100)
    with contextlib.redirect_stdout(io.StringIO()):                          # stray print at IPDnet2.py:149
        l0, l1 = net.layers[0], net.layers[1]
        x0 = T(rs_randn(2101, (2, 32, 6, 96)))                               # [B, F, T, H]
        arrs["fconv1_out"] = l0._fconv(l0.fconv1, x0).numpy()
        x128 = T(rs_randn(2102, (1, 128, 3, 96)))
        arrs["full128_out"] = l0._full(x128).numpy()
        x16 = T(rs_randn(2103, (2, 16, 5, 96)))
        arrs["full16_out"] = l1._full(x16).numpy()
        arrs["fconv2_l1_out"] = l1._fconv(l1.fconv2, x16).numpy()
        arrs["pool2_out"] = l0.fre_compress_first(x0.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).numpy()
        arrs["pool8_out"] = l0.fre_compress_second(x0.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).numpy()
        xt = T(rs_randn(2104, (6, 13, 96)))
        arrs["tpool_out"] = net.time_pooling(xt).numpy()
        xf = T(rs_randn(2105, (2, 96, 4, 16)))                               # [B, H, T', Fc]
        arrs["finv_out"] = net.freq_inverse(xf).numpy()
        # layer orchestration (Mamba = the restatement above)
        xl = T(rs_randn(2106, (1, 16, 10, 96)))
        arrs["layer1_out"] = l1(xl, None, True, False, None, False)[0].numpy()
        xl0 = T(rs_randn(2107, (1, 256, 10, 96), 0.5))
        arrs["layer0_out"] = l0(xl0, None, True, False, None, False)[0].numpy()
        arrs["mamba_out"] = l1.mhsa(T(rs_randn(2108, (3, 17, 96)))).numpy()   # the restatement itself (unpinned)
        # whole network, 5-mic config of run_IPDnet2.py:103-119
        arrs["net_out"] = net(T(rs_randn(2110, (2, 10, 256, 20)))).numpy()
    # 15-mic input (BASELINE config 5 mapping: dim_input = 30), 3 layers to keep it small
    sd3, net3 = build(2200, dim_input=30, num_layers=3)
    with contextlib.redirect_stdout(io.StringIO()):
        arrs["net30_out"] = net3(T(rs_randn(2210, (1, 30, 256, 15)))).numpy()
    path = os.path.join(HERE, "g14_ipdnet2.npz")
    np.savez_compressed(path, **arrs)
    print("g14_ipdnet2 %.1f KB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
