"""Drop-in for the reference's ``IPDnet2/IPDnet2.py`` (online SpatialNet with Mamba blocks): same classes,
constructor arguments, ``forward`` signatures, return shapes and ``state_dict`` key names, with every ``forward``
running the HIP kernels of libfnssl_hip.so (``fnssl_sn_*``) on an MI355X.

    LayerNorm(seq_last, normalized_shape=...)                              arch/base/norm.py:11-27
    CausalConv1d(in, out, k, look_ahead=0).forward(x[B, C, T], state=None)  IPDnet2.py:45-82
    FreqInverse(nfreq, compression_ratio, hidden_dim, out_dim).forward(x[B, H, T, Fc]) -> [B, out, T, nfreq]   :23-43
    Mamba(d_model, d_state, d_conv)       parameter holder with mamba_ssm's names (the package is not needed)
    SpatialNetLayer(...).forward(x[B, F, T, H], ...) -> (x, None)           :85-164
    OnlineSpatialNet(...).forward(x[B, dim_input, F, T], inference=False) -> [B, T//5, 2F, dim_output//4, 2]   :259-368
        .forward_stream(x_chunk, state=None) -> (out, state)   streaming with carried state (chunks of 5 k frames)

Built for the configuration the reference ships (run_IPDnet2.py:103-119): dim_hidden 96, dim_squeeze 8,
kernel_size (5, .), conv_groups (8, .), all-LN norms, attention 'mamba(16,4)', dim_output 16,
fre_compression_ratio 16, time_compression_layer 0; other values raise.  Forward only (``eval()``), fp32, ROCm
tensors only — there is no CPU path.  ``inference=True`` (the reference's frame-by-frame Mamba stepping, :170-177)
is honoured: the Mamba blocks are driven one frame per ``fnssl_sn_mamba`` call with carried conv / SSM state (the
role of ``InferenceParams``); it computes the same function as the parallel mode with T x the launches.  The Mamba block follows the
published algorithm ("parity unpinned": no mamba_ssm to compare with, see oracle/ipdnet2_oracle.py).
"""
import math
import os
import sys
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from fnssl import ops                                               # noqa: E402
from fnssl import spatialnet as sn                                  # noqa: E402
from Model import _is_bf16, _param_key, _require_eval               # noqa: E402


def _prec(module) -> int:
    """After ``module.bfloat16()`` the dense products (encoder, grouped f-conv, full-band branch, Mamba in / x / out
    projections) run on bf16 MFMAs with fp32 accumulation and fp32 tensors — BASELINE config 5 as written
    (include/fnssl.h)."""
    return sn.BF16 if _is_bf16(module) else sn.FP32


def _sd(module, prefix=""):
    return {prefix + k: v for k, v in module.state_dict().items()}


class LayerNorm(nn.LayerNorm):
    """arch/base/norm.py:11-27: nn.LayerNorm over H, optionally with the sequence dim last."""

    def __init__(self, seq_last: bool, **kwargs) -> None:
        super().__init__(**kwargs)
        self.seq_last = seq_last

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if self.seq_last:
            input = input.transpose(-1, 1)
        o = sn.layernorm(input.float(), self.weight.float(), self.bias.float(), self.eps).to(input.dtype)
        if self.seq_last:
            o = o.transpose(-1, 1)
        return o


def new_norm(norm_type: str, dim_hidden: int, seq_last: bool, group_size: int = None, num_groups: int = None,
             dims_norm: List[int] = None, dim_affine: int = None) -> nn.Module:
    """arch/base/norm.py:232-247; only 'LN' is on the shipped path."""
    if norm_type.upper() == 'LN':
        return LayerNorm(normalized_shape=dim_hidden, seq_last=seq_last)
    raise NotImplementedError("norm %r: the MI355X path implements the shipped all-LN configuration" % norm_type)


class FreqInverse(nn.Module):
    def __init__(self, nfreq=256, compression_ratio=16, hidden_dim=96, out_dim=16, sample_rate=16000):
        super().__init__()
        self.nfreq = nfreq
        self.nfilters = nfreq // compression_ratio
        self.sample_rate = sample_rate
        self.hidden_dim = hidden_dim
        self.out_dim = out_dim
        self.compression_ratio = compression_ratio
        self.trans2 = nn.Conv1d(self.hidden_dim, compression_ratio * self.out_dim, 1)

    def forward(self, x):
        """x [B, H, T, Fc] -> tanh(scatter(trans2(x))) [B, out_dim, T, nfreq]   (:37-43).  Runs the fused head
        kernel with an identity decoder and un-does its output ordering (views only)."""
        if self.compression_ratio != 16 or self.out_dim != 16 or self.hidden_dim != 96:
            raise RuntimeError("FreqInverse: built for compression 16, out_dim 16, hidden 96")
        dev = x.device
        sd = {"freq_inverse." + k: v for k, v in self.state_dict().items()}
        sd["decoder.weight"] = torch.eye(16)
        sd["decoder.bias"] = torch.zeros(16)
        ptrs, keep = sn.pack_head(sd, dev)
        B, H, T, Fc = x.shape
        y = sn.head(x.float().permute(0, 3, 2, 1), ptrs)             # [B, T, 2F, 4, 2]; flat (f, gg, m, a)
        y = y.reshape(B, T, Fc * 16, 2, 4, 2).permute(0, 5, 3, 4, 1, 2)   # [B, a, gg, m, T, F]; o = a*8 + gg*4 + m
        del keep
        return y.reshape(B, 16, T, Fc * 16).to(x.dtype)


class CausalConv1d(nn.Conv1d):
    """IPDnet2.py:45-82.  The HIP kernel is the network's encoder: out_channels 96, kernel 5, look_ahead 0."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, padding_mode: str = 'zeros', device=None, dtype=None,
                 look_ahead: int = 0) -> None:
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode,
                         device, dtype)
        self.look_ahead = look_ahead
        assert look_ahead <= self.kernel_size[0] - 1, (look_ahead, self.kernel_size)

    def _packed(self, device):
        key = (_param_key(self), str(device))
        if getattr(self, "_pk", None) != key:
            if (self.out_channels, self.kernel_size[0], self.look_ahead, self.groups) != (96, 5, 0, 1) or self.bias is None:
                raise RuntimeError("CausalConv1d: the MI355X kernel is the encoder conv (out 96, k 5, look_ahead 0, bias)")
            self._wT = self.weight.detach().float().to(device).permute(1, 2, 0).contiguous()
            self._b = self.bias.detach().float().to(device).contiguous()
            self._pk = key
        return self._wT, self._b

    def forward(self, x: torch.Tensor, state: Dict[int, Any] = None) -> torch.Tensor:
        """x [B, H, T] -> [B, 96, T]; ``state`` dict as in the reference (:69-74; the reference's own state branch
        cannot run — ``-self.kernel_size`` negates a tuple — so only its contract is kept): state[id(self)] holds
        the last 4 input frames [B, H, 4]."""
        wT, b = self._packed(x.device)
        x4 = x.float().unsqueeze(2)                                   # [B, C, F = 1, T]
        st_in = st_out = None
        if state is not None:
            if id(self) in state:
                st_in = state[id(self)].float().contiguous().unsqueeze(2)
            st_out = torch.empty((x.shape[0], x.shape[1], 1, 4), dtype=torch.float32, device=x.device)
        y = sn.encoder(x4, wT, b, st_in, st_out, precision=_prec(self))   # logical [B, 1, T, 96]
        if state is not None:
            state[id(self)] = st_out[:, :, 0, :]
        return y[:, 0].transpose(1, 2).to(x.dtype)


class Mamba(nn.Module):
    """Parameter holder with the names / shapes of ``mamba_ssm.Mamba(d_model, d_state, d_conv, expand=2)``
    (reference call sites IPDnet2.py:127,132), so reference checkpoints load.  Compute = fnssl_sn_mamba."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, layer_idx=None):
        super().__init__()
        self.d_model, self.d_state, self.d_conv = d_model, d_state, d_conv
        self.d_inner = expand * d_model
        self.dt_rank = math.ceil(d_model / 16)
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=False)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1, bias=True)
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True)
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(self.d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=False)


class SpatialNetLayer(nn.Module):

    def __init__(self, dim_hidden: int, dim_squeeze: int, num_freqs: int, dropout: Tuple[float, float, float] = (0, 0, 0),
                 kernel_size: Tuple[int, int] = (5, 3), conv_groups: Tuple[int, int] = (8, 8),
                 norms: List[str] = ["LN", "LN", "GN", "LN", "LN", "LN"], padding: str = 'zeros', full: nn.Module = None,
                 attention: str = 'mhsa', is_first: bool = False) -> None:
        super().__init__()
        if not attention.startswith("mamba("):
            raise NotImplementedError("attention %r: the MI355X path implements the shipped 'mamba(16,4)'" % attention)
        if (dim_hidden, dim_squeeze, kernel_size[0], conv_groups[0], padding) != (96, 8, 5, 8, 'zeros') or any(dropout):
            raise NotImplementedError("SpatialNetLayer: built for dim_hidden 96, dim_squeeze 8, f-kernel 5, 8 groups, "
                                      "zero padding, no dropout")
        f_conv_groups, t_conv_groups = conv_groups
        f_kernel_size = kernel_size[0]
        self.fconv1 = nn.ModuleList([
            new_norm(norms[3], dim_hidden, seq_last=True, group_size=None, num_groups=f_conv_groups),
            nn.Conv1d(dim_hidden, dim_hidden, f_kernel_size, groups=f_conv_groups, padding='same', padding_mode=padding),
            nn.PReLU(dim_hidden),
        ])
        self.norm_full = new_norm(norms[5], dim_hidden, seq_last=False, group_size=None, num_groups=f_conv_groups)
        self.full_share = False if full is None else True
        self.dim_squeeze = dim_squeeze
        self.squeeze = nn.Sequential(nn.Conv1d(dim_hidden, dim_squeeze, 1), nn.SiLU())
        self.dropout_full = None
        self.is_first = is_first
        self.full = nn.Linear(num_freqs, num_freqs) if full is None else full
        self.unsqueeze = nn.Sequential(nn.Conv1d(dim_squeeze, dim_hidden, 1), nn.SiLU())
        self.fconv2 = nn.ModuleList([
            new_norm(norms[4], dim_hidden, seq_last=True, group_size=None, num_groups=f_conv_groups),
            nn.Conv1d(dim_hidden, dim_hidden, f_kernel_size, groups=f_conv_groups, padding='same', padding_mode=padding),
            nn.PReLU(dim_hidden),
        ])
        self.norm_mhsa = new_norm(norms[0], dim_hidden, seq_last=False, group_size=None, num_groups=t_conv_groups)
        attn_params = attention[6:-1].split(',')
        d_state, mamba_conv_kernel = int(attn_params[0]), int(attn_params[1])
        if (d_state, mamba_conv_kernel) != (16, 4):
            raise NotImplementedError("SpatialNetLayer: built for mamba(16,4)")
        self.mhsa = Mamba(d_model=dim_hidden, d_state=d_state, d_conv=mamba_conv_kernel, layer_idx=0)
        self.attention = attention
        self.dropout_mhsa = nn.Dropout(dropout[0])
        self.norm_tconvffn = new_norm(norms[1], dim_hidden, seq_last=False, group_size=None, num_groups=t_conv_groups)
        self.tconvffn = Mamba(d_model=dim_hidden, d_state=d_state, d_conv=mamba_conv_kernel, layer_idx=0)
        self.dropout_tconvffn = nn.Dropout(dropout[1])
        self.fre_compress_second = nn.AvgPool2d(kernel_size=(1, 8))
        self.fre_compress_first = nn.AvgPool2d(kernel_size=(1, 2))
        self._pk = None

    def _packed(self, device):
        key = (_param_key(self), str(device))
        if self._pk != key:
            sd = _sd(self)
            keep = sn._Keep()
            self._w = (sn.pack_fconv(sd, "fconv1", device, keep)[0], sn.pack_full(sd, "", device, keep)[0],
                       sn.pack_fconv(sd, "fconv2", device, keep)[0],
                       sn.pack_mamba(sd, "norm_mhsa", "mhsa", device, keep)[0],
                       sn.pack_mamba(sd, "norm_tconvffn", "tconvffn", device, keep)[0])
            self._keep, self._pk = keep, key
        return self._w

    # the three branches on their own (WITHOUT the residual), as the reference's private helpers return them
    def _fconv(self, ml: nn.ModuleList, x: torch.Tensor) -> torch.Tensor:
        w = self._packed(x.device)
        return sn.fconv(x.float(), w[0] if ml is self.fconv1 else w[2], residual=False, precision=_prec(self)).to(x.dtype)

    def _full(self, x: torch.Tensor) -> torch.Tensor:
        return sn.full(x.float(), self._packed(x.device)[1], residual=False, precision=_prec(self)).to(x.dtype)

    def _mamba(self, x: torch.Tensor, mamba: Mamba, norm: nn.Module, dropout: nn.Module, inference: bool = False):
        """LN + Mamba along T WITHOUT the residual (:166-181).  ``inference``: frame-by-frame stepping from zero state,
        the carried conv taps / SSM state playing ``InferenceParams`` (:170-177)."""
        w = self._packed(x.device)
        mw = w[3] if mamba is self.mhsa else w[4]
        xf = x.float()
        if not inference:
            return sn.mamba(xf, mw, residual=False, precision=_prec(self)).to(x.dtype)
        return self._mamba_steps(xf, mw, False).to(x.dtype)

    def _mamba_steps(self, xf: torch.Tensor, mw, residual: bool) -> torch.Tensor:
        nb, nf, nt, _ = xf.shape
        st = sn.mamba_state(nb, nf, xf.device)
        out = sn._new_bfth(nb, nf, nt, xf.device)
        for t in range(nt):
            sn.mamba(xf[:, :, t:t + 1], mw, residual=residual, state=st, carry=t > 0, out=out[:, :, t:t + 1],
                     precision=_prec(self))
        return out

    @ops.on_device
    def forward(self, x: torch.Tensor, att_mask: Optional[torch.Tensor] = None, chunkwise_recurrent: bool = True,
                rope: bool = True, state: Dict[int, Any] = None, inference: bool = False):
        """x [B, F, T, H] -> (x, None)   (:137-164; the first layer shrinks F by 2 and then by 8)."""
        _require_eval(self)
        f1, fu, f2, m0, m1 = self._packed(x.device)
        y = x.float()
        pr = _prec(self)
        y = sn.fconv(y, f1, pool=2 if self.is_first else 1, precision=pr)   # x + fconv1, fre_compress_first
        y = sn.full(y, fu, out=y, precision=pr)
        y = sn.fconv(y, f2, pool=8 if self.is_first else 1, precision=pr)   # x + fconv2, fre_compress_second
        if inference:                                                       # per-frame recurrence (:170-177)
            y = self._mamba_steps(self._mamba_steps(y, m0, True), m1, True)
        else:
            y = sn.mamba(y, m0, out=y, precision=pr)
            y = sn.mamba(y, m1, out=y, precision=pr)
        return y.to(x.dtype), None

    def extra_repr(self) -> str:
        return f"full_share={self.full_share}"


class OnlineSpatialNet(nn.Module):

    def __init__(self, dim_input: int, dim_output: int, num_layers: int, dim_squeeze: int, num_freqs: int,
                 encoder_kernel_size: int = 5, dim_hidden: int = 192, num_heads: int = 2,
                 dropout: Tuple[float, float, float] = (0, 0, 0), kernel_size: Tuple[int, int] = (5, 3),
                 conv_groups: Tuple[int, int] = (8, 8), norms: List[str] = ["LN", "LN", "GN", "LN", "LN", "LN"],
                 padding: str = 'zeros', attention: str = 'mhsa(251)', chunkwise_recurrent: bool = True,
                 rope: Union[bool, str] = False, fre_compression_ratio: int = 16, time_compression_ratio: int = 5,
                 time_compression_layer: int = 0):
        super().__init__()
        if (dim_output, encoder_kernel_size, fre_compression_ratio, time_compression_layer) != (16, 5, 16, 0):
            raise NotImplementedError("OnlineSpatialNet: built for dim_output 16, encoder kernel 5, frequency "
                                      "compression 16, time compression in layer 0 (run_IPDnet2.py:103-119)")
        self.num_heads = num_heads
        self.chunkwise_recurrent = chunkwise_recurrent
        self.pos = None
        self.attn_scope = 1
        self.rope = rope
        self.encoder = CausalConv1d(in_channels=dim_input, out_channels=dim_hidden, kernel_size=encoder_kernel_size,
                                    look_ahead=0)
        self.time_compression_layer = time_compression_layer
        self.time_compression_ratio = time_compression_ratio
        self.num_freqs = num_freqs
        layers = []
        for l in range(num_layers):
            layers.append(SpatialNetLayer(
                dim_hidden=dim_hidden, dim_squeeze=dim_squeeze,
                num_freqs=num_freqs // 2 if l == 0 else num_freqs // fre_compression_ratio,
                dropout=dropout, kernel_size=kernel_size, conv_groups=conv_groups, norms=norms, padding=padding, full=None,
                attention=attention, is_first=(l == 0)))
        self.layers = nn.ModuleList(layers)
        self.freq_inverse = FreqInverse(nfreq=num_freqs, compression_ratio=fre_compression_ratio, hidden_dim=dim_hidden,
                                        out_dim=dim_output)
        self.decoder = nn.Linear(in_features=dim_output, out_features=dim_output)
        self.time_pooling = nn.AvgPool2d(kernel_size=(time_compression_ratio, 1))
        self._dn = None
        self._dn_key = None

    def device_net(self, device) -> "sn.DeviceSpatialNet":
        key = (_param_key(self), str(device))
        if self._dn is None or self._dn_key != key:
            self._dn = sn.DeviceSpatialNet(self.state_dict(), device, time_ratio=self.time_compression_ratio,
                                           precision=_prec(self))
            self._dn_key = key
        return self._dn

    @ops.on_device
    def forward(self, x: torch.Tensor, inference: bool = False, return_attn_score: bool = False):
        """x [B, dim_input, F, T] -> [B, T // 5, 2F, 4, 2]   (:331-368)."""
        _require_eval(self)
        if return_attn_score:
            raise NotImplementedError("return_attn_score: the Mamba configuration has no attention scores")
        dn = self.device_net(x.device)
        if inference:       # the reference's frame-by-frame Mamba stepping (:170-177), see DeviceSpatialNet.forward_stepwise
            return dn.forward_stepwise(x.float()).to(x.dtype)
        return dn.forward(x.float()).to(x.dtype)

    @ops.on_device
    def forward_stream(self, x: torch.Tensor, state=None):
        """The online / causal path with carried state: x = the NEXT T frames (T a positive multiple of the time
        compression ratio); ``state`` is None for the first chunk, afterwards what the previous call returned.
        Consecutive chunks reproduce ``forward`` on the whole signal (encoder taps, the Mamba conv taps and the
        SSM states are carried; every other op is local in time)."""
        _require_eval(self)
        if x.shape[3] == 0 or x.shape[3] % self.time_compression_ratio:
            raise RuntimeError("OnlineSpatialNet.forward_stream: chunks must be positive multiples of %d frames, got %d"
                               % (self.time_compression_ratio, x.shape[3]))
        dn = self.device_net(x.device)
        carry = state is not None
        if state is None:
            state = dn.new_state(x.shape[0])
        out = dn.forward(x.float(), state=state, carry=carry)
        return out.to(x.dtype), state
