"""Torch-tensor front of the IPDnet2 / OnlineSpatialNet entry points of libfnssl_hip.so (``fnssl_sn_*``,
include/fnssl.h; reference IPDnet2/IPDnet2.py).  PyTorch supplies device memory and the current HIP stream;
every numeric op is a HIP kernel.  No CPU path: non-ROCm tensors raise.

Activations are logical ``[B, F, T, H]`` tensors (the reference's layout) with ANY strides as long as H is
contiguous; internally the library addresses them as [B, T, F, H] views.  Weights are re-laid-out once per model
(``pack_*``): the kernels read them as wave-uniform scalars, transposed to [in][out].
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BtfView, SnFconvW, SnFullW, SnMambaW, SnNet, check
from .ops import _need_dev, _ptr, _stream, _workspace, on_device

H, HS, E, NST, RK, KC, XP, DO = 96, 8, 192, 16, 6, 4, 40, 16
FP32, BF16 = 0, 1                      # FNSSL_PRECISION_FP32 / _BF16


def _t(a, device):
    t = a.detach() if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _view(x):
    """Logical [B, F, T, H] tensor (any strides, H contiguous) -> BtfView (sb, st, sf)."""
    sb, sf, st, sh = x.stride()
    if sh != 1 or x.shape[3] != H:
        raise RuntimeError("fnssl.spatialnet: expected [B, F, T, %d] with contiguous channels, got %s strides %s"
                           % (H, tuple(x.shape), x.stride()))
    return BtfView(x.data_ptr(), sb, st, sf)


def _conform(x):
    ok = x.stride(-1) == 1 and x.data_ptr() % 16 == 0 and all(s % 4 == 0 for s in x.stride()[:-1])
    return x if ok else x.contiguous()


def _new_bfth(nb, nf, nt, device):
    """A logical [B, F, T, H] tensor stored frame-major ([B, T, F, H] in memory), the library's native layout."""
    return torch.empty((nb, nt, nf, H), dtype=torch.float32, device=device).permute(0, 2, 1, 3)


# --------------------------------------------------------------------------------------------------------- #
# weight packing (state_dict names of the reference / of mamba_ssm.Mamba)
# --------------------------------------------------------------------------------------------------------- #
class _Keep:
    """Holds the device tensors a ctypes weight struct points to."""

    def __init__(self):
        self.tensors = []

    def add(self, t):
        self.tensors.append(t)
        return t.data_ptr()


def _check_shape(name, t, shape):
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError("fnssl.spatialnet: %s has shape %s, this build needs %s (dim_hidden 96, dim_squeeze 8, "
                           "f-conv k5/g8, mamba(16,4), dim_output 16)" % (name, tuple(t.shape), tuple(shape)))


def pack_fconv(sd, prefix, device, keep=None):
    """prefix + '.0' LayerNorm, '.1' Conv1d(96, 96, 5, groups 8), '.2' PReLU(96)   (IPDnet2.py:105-109)."""
    keep = keep or _Keep()
    w = _t(sd[prefix + ".1.weight"], device)
    _check_shape(prefix + ".1.weight", w, (H, 12, 5))
    wT = w.view(8, 12, 12, 5).permute(0, 3, 2, 1).contiguous()          # [g][tap][ci][o]
    s = SnFconvW(keep.add(_t(sd[prefix + ".0.weight"], device)), keep.add(_t(sd[prefix + ".0.bias"], device)),
                 keep.add(wT), keep.add(_t(sd[prefix + ".1.bias"], device)),
                 keep.add(_t(sd[prefix + ".2.weight"], device)))
    return s, keep


def pack_full(sd, prefix, device, keep=None):
    """prefix + 'norm_full', 'squeeze.0', 'full', 'unsqueeze.0'   (IPDnet2.py:111-118)."""
    keep = keep or _Keep()
    ws = _t(sd[prefix + "squeeze.0.weight"], device)
    _check_shape(prefix + "squeeze.0.weight", ws, (HS, H, 1))
    wu = _t(sd[prefix + "unsqueeze.0.weight"], device)
    wf = _t(sd[prefix + "full.weight"], device)
    s = SnFullW(keep.add(_t(sd[prefix + "norm_full.weight"], device)), keep.add(_t(sd[prefix + "norm_full.bias"], device)),
                keep.add(ws[:, :, 0].t().contiguous()), keep.add(_t(sd[prefix + "squeeze.0.bias"], device)),
                keep.add(wf.t().contiguous()), keep.add(_t(sd[prefix + "full.bias"], device)),
                keep.add(wu[:, :, 0].t().contiguous()), keep.add(_t(sd[prefix + "unsqueeze.0.bias"], device)))
    return s, keep, wf.shape[0]


def pack_mamba(sd, p_norm, p_mamba, device, keep=None):
    """LayerNorm p_norm + mamba_ssm.Mamba(96, d_state 16, d_conv 4) parameters under p_mamba (IPDnet2.py:126-132)."""
    keep = keep or _Keep()
    g = lambda n: _t(sd[p_mamba + "." + n], device)   # noqa: E731
    win = g("in_proj.weight")
    _check_shape(p_mamba + ".in_proj.weight", win, (2 * E, H))
    wx = g("x_proj.weight")
    _check_shape(p_mamba + ".x_proj.weight", wx, (RK + 2 * NST, E))
    wxT = torch.zeros((E, XP), dtype=torch.float32, device=device)
    wxT[:, :RK + 2 * NST] = wx.t()
    cw = g("conv1d.weight")
    _check_shape(p_mamba + ".conv1d.weight", cw, (E, 1, KC))
    s = SnMambaW(keep.add(_t(sd[p_norm + ".weight"], device)), keep.add(_t(sd[p_norm + ".bias"], device)),
                 keep.add(win.t().contiguous()), keep.add(cw[:, 0, :].contiguous()), keep.add(g("conv1d.bias")),
                 keep.add(wxT), keep.add(g("dt_proj.weight")), keep.add(g("dt_proj.bias")),
                 keep.add((-torch.exp(g("A_log"))).contiguous()), keep.add(g("D")),
                 keep.add(g("out_proj.weight").t().contiguous()))
    return s, keep


def pack_head(sd, device, keep=None):
    keep = keep or _Keep()
    w = _t(sd["freq_inverse.trans2.weight"], device)
    _check_shape("freq_inverse.trans2.weight", w, (16 * DO, H, 1))
    wfiP = w[:, :, 0].view(DO, 16, H).permute(1, 0, 2).contiguous()     # [r][o][h]
    bfiP = _t(sd["freq_inverse.trans2.bias"], device).view(DO, 16).t().contiguous()
    wd = _t(sd["decoder.weight"], device)
    _check_shape("decoder.weight", wd, (DO, DO))
    ptrs = (keep.add(wfiP), keep.add(bfiP), keep.add(wd.t().contiguous()), keep.add(_t(sd["decoder.bias"], device)))
    return ptrs, keep


# --------------------------------------------------------------------------------------------------------- #
# single ops (each = one C entry point)
# --------------------------------------------------------------------------------------------------------- #
@on_device
def layernorm(x, w, b, eps: float = 1e-5):
    """LayerNorm over the last dim of a contiguous tensor (arch/base/norm.py:11-27), wavefront reductions."""
    _need_dev(x, w, b)
    x = x.contiguous()
    h = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.load().fnssl_sn_layernorm(_ptr(x), x.numel() // h, h, _ptr(w.contiguous()), _ptr(b.contiguous()), eps,
                                         _ptr(y), _stream()), "sn_layernorm")
    return y


@on_device
def encoder(x, wT, bias, state_in=None, state_out=None, precision: int = FP32):
    """x [B, C, F, T] (any strides) -> logical [B, F, T, 96]; wT [C][5][96].  state_*: [B, C, F, 4] or None.
    ``precision``: FP32, or BF16 = bf16 MFMA operands, fp32 accumulate (include/fnssl.h, fnssl_sn_encoder)."""
    _need_dev(x, wT, bias, state_in, state_out)
    nb, cin, nf, nt = x.shape
    out = _new_bfth(nb, nf, nt, x.device)
    v = _view(out)
    sb, sc, sf, st = x.stride()
    check(_lib.load().fnssl_sn_encoder(_ptr(x), sb, sc, sf, st, nb, cin, nf, nt, _ptr(wT), _ptr(bias),
                                       _ptr(state_in), _ptr(state_out), v.p, v.sb, v.st, v.sf, int(precision), _stream()),
          "sn_encoder")
    return out


@on_device
def fconv(x, w: SnFconvW, residual: bool = True, pool: int = 1, out=None, precision: int = FP32):
    """x + PReLU(Conv_g(LN(x))) along F (+ AvgPool over F).  x logical [B, F, T, 96]; returns logical
    [B, F // pool, T, 96].  ``out`` may be x itself when pool == 1."""
    _need_dev(x, out)
    x = _conform(x)
    nb, nf, nt, _ = x.shape
    if out is None:
        out = _new_bfth(nb, nf // pool, nt, x.device)
    xv, ov = _view(x), _view(out)
    check(_lib.load().fnssl_sn_fconv(C.byref(xv), nb, nt, nf, C.byref(w), int(residual), pool, ov.p, ov.sb, ov.st, ov.sf,
                                     int(precision), _stream()), "sn_fconv")
    return out


@on_device
def full(x, w: SnFullW, residual: bool = True, out=None, precision: int = FP32):
    """x + SiLU(unsqueeze(Linear_F(SiLU(squeeze(LN(x))))))   (IPDnet2.py:235-253)."""
    _need_dev(x, out)
    x = _conform(x)
    nb, nf, nt, _ = x.shape
    if out is None:
        out = _new_bfth(nb, nf, nt, x.device)
    xv, ov = _view(x), _view(out)
    check(_lib.load().fnssl_sn_full(C.byref(xv), nb, nt, nf, C.byref(w), int(residual), ov.p, ov.sb, ov.st, ov.sf,
                                    int(precision), _stream()), "sn_full")
    return out


def mamba_state(nb: int, nf: int, device):
    """(conv_state [nb*nf, 3, 192], ssm_state [nb*nf, 192, 16]) zero-initialised."""
    return (torch.zeros((nb * nf, KC - 1, E), dtype=torch.float32, device=device),
            torch.zeros((nb * nf, E, NST), dtype=torch.float32, device=device))


@on_device
def mamba(x, w: SnMambaW, residual: bool = True, time_pool: int = 1, state=None, carry: bool = False, out=None,
          precision: int = FP32):
    """x + Mamba(LN(x)) along T for every (b, f) (+ AvgPool over T).  state = mamba_state(...) or None."""
    _need_dev(x, out)
    x = _conform(x)
    nb, nf, nt, _ = x.shape
    if out is None:
        out = _new_bfth(nb, nf, nt // time_pool, x.device)
    lib = _lib.load()
    ws = _workspace(lib.fnssl_sn_mamba_workspace_bytes(nb, nt, nf), x.device, "sn_mamba")
    xv, ov = _view(x), _view(out)
    cs, ss = state if state is not None else (None, None)
    _need_dev(cs, ss)
    check(lib.fnssl_sn_mamba(C.byref(xv), nb, nt, nf, C.byref(w), int(residual), time_pool, _ptr(cs), _ptr(ss),
                             int(carry), ov.p, ov.sb, ov.st, ov.sf, _ptr(ws), ws.numel(), int(precision), _stream()), "sn_mamba")
    return out


@on_device
def head(x, head_ptrs):
    """x logical [B, Fc, T', 96] -> [B, T', 2*16*Fc, 4, 2]  (FreqInverse + tanh + decoder + re-ordering)."""
    _need_dev(x)
    x = _conform(x)
    nb, nfc, nt2, _ = x.shape
    out = torch.empty((nb, nt2, 2 * 16 * nfc, DO // 4, 2), dtype=torch.float32, device=x.device)
    xv = _view(x)
    check(_lib.load().fnssl_sn_head(C.byref(xv), nb, nt2, nfc, *[C.c_void_p(p) for p in head_ptrs], _ptr(out), _stream()),
          "sn_head")
    return out


# --------------------------------------------------------------------------------------------------------- #
# whole network
# --------------------------------------------------------------------------------------------------------- #
class DeviceSpatialNet:
    """Device-resident re-laid-out parameters of an OnlineSpatialNet (the ``fnssl_sn_net`` struct)."""

    def __init__(self, state: dict, device, prefix: str = "", time_ratio: int = 5, precision: int = FP32):
        self.device = torch.device(device)
        self.precision = int(precision)
        sd = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
        self.keep = _Keep()
        net = SnNet()
        enc = _t(sd["encoder.weight"], self.device)
        if enc.shape[0] != H or enc.shape[2] != 5:
            raise RuntimeError("fnssl.spatialnet: encoder.weight %s, this build needs [96, dim_input, 5]" % (tuple(enc.shape),))
        self.dim_input = enc.shape[1]
        self.num_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
        if self.num_layers > _lib.SN_MAX_LAYERS:
            raise RuntimeError("fnssl.spatialnet: at most %d layers" % _lib.SN_MAX_LAYERS)
        net.dim_input, net.num_layers, net.time_ratio = self.dim_input, self.num_layers, int(time_ratio)
        net.precision = self.precision
        self.enc_wT = enc.permute(1, 2, 0).contiguous()                 # [c][k][o]
        self.enc_b = _t(sd["encoder.bias"], self.device)
        net.enc_wT, net.enc_b = self.keep.add(self.enc_wT), self.keep.add(self.enc_b)
        self.layers = []
        for l in range(self.num_layers):
            p = "layers.%d." % l
            f1, _ = pack_fconv(sd, p + "fconv1", self.device, self.keep)
            f2, _ = pack_fconv(sd, p + "fconv2", self.device, self.keep)
            fu, _, nfull = pack_full(sd, p, self.device, self.keep)
            m0, _ = pack_mamba(sd, p + "norm_mhsa", p + "mhsa", self.device, self.keep)
            m1, _ = pack_mamba(sd, p + "norm_tconvffn", p + "tconvffn", self.device, self.keep)
            L = net.layers[l]
            L.fconv1, L.fconv2, L.full = f1, f2, fu
            L.mamba[0], L.mamba[1] = m0, m1
            self.layers.append((f1, fu, f2, m0, m1, nfull))
        self.num_freqs = 2 * self.layers[0][5]
        if self.num_layers > 1 and self.layers[1][5] * 16 != self.num_freqs:
            raise RuntimeError("fnssl.spatialnet: full.weight sizes %d / %d do not describe a 2 x 8 frequency compression"
                               % (self.layers[0][5], self.layers[1][5]))
        self.head_ptrs, _ = pack_head(sd, self.device, self.keep)
        net.wfiP, net.bfiP, net.wdT, net.bd = self.head_ptrs
        self.net = net
        self.time_ratio = int(time_ratio)

    def forward_stepwise(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's ``inference=True`` (IPDnet2.py:170-177): every Mamba block is driven FRAME BY FRAME from zero
        state — one ``fnssl_sn_mamba`` call per frame with the carried conv taps / SSM state standing in for
        ``InferenceParams`` — and the network is orchestrated per op (fnssl_sn_encoder / fconv / full / mamba / head,
        fnssl_avgpool_time for the 5x time pooling) instead of by ``fnssl_sn_forward``.  Same function as ``forward``
        (the published recurrence and its scan are one algorithm); T x more launches, so it is the faithful twin of the
        flag, not the fast path."""
        from . import ops
        _need_dev(x)
        with torch.cuda.device(x.device):
            nb, cin, nf, nt = x.shape
            if cin != self.dim_input or nf != self.num_freqs:
                raise RuntimeError("fnssl.spatialnet: expected [B, %d, %d, T], got %s"
                                   % (self.dim_input, self.num_freqs, tuple(x.shape)))
            pr = self.precision
            y = encoder(x, self.enc_wT, self.enc_b, precision=pr)                  # logical [B, F, T, 96]
            for l, (f1, fu, f2, m0, m1, _) in enumerate(self.layers):
                first = l == 0
                y = fconv(y, f1, pool=2 if first else 1, precision=pr)
                y = full(y, fu, out=y, precision=pr)
                y = fconv(y, f2, pool=8 if first else 1, precision=pr)
                for m in (m0, m1):
                    st = mamba_state(nb, y.shape[1], x.device)
                    out = _new_bfth(nb, y.shape[1], y.shape[2], x.device)
                    for t in range(y.shape[2]):
                        mamba(y[:, :, t:t + 1], m, residual=True, state=st, carry=t > 0, out=out[:, :, t:t + 1], precision=pr)
                    y = out
                if first:
                    y = ops.avgpool_time(y.contiguous(), self.time_ratio)         # [B, Fc, T // ratio, 96]
            return head(y, self.head_ptrs)

    def new_state(self, nb: int):
        n = _lib.load().fnssl_sn_state_floats(C.byref(self.net), nb, self.num_freqs)
        return torch.zeros(n, dtype=torch.float32, device=self.device)

    def forward(self, x: torch.Tensor, state=None, carry: bool = False) -> torch.Tensor:
        """x [B, dim_input, F, T] (any strides) -> [B, T // 5, 2F, 4, 2].  ``state``: a ``new_state`` tensor that is
        updated in place (streaming; T must then be a multiple of 5), read when ``carry``."""
        _need_dev(x, state)
        with torch.cuda.device(x.device):
            nb, cin, nf, nt = x.shape
            if cin != self.dim_input or nf != self.num_freqs:
                raise RuntimeError("fnssl.spatialnet: expected [B, %d, %d, T], got %s"
                                   % (self.dim_input, self.num_freqs, tuple(x.shape)))
            lib = _lib.load()
            ws = _workspace(lib.fnssl_sn_forward_workspace_bytes(nb, nf, nt), x.device, "sn_forward")
            out = torch.empty((nb, nt // self.time_ratio, 2 * nf, DO // 4, 2), dtype=torch.float32, device=x.device)
            sb, sc, sf, st = x.stride()
            check(lib.fnssl_sn_forward(C.byref(self.net), _ptr(x), sb, sc, sf, st, nb, nf, nt, _ptr(state), int(carry),
                                       _ptr(out), _ptr(ws), ws.numel(), _stream()), "sn_forward")
            return out
