"""CPU oracle for the IPDnet2 row (OnlineSpatialNet).  TEST INFRASTRUCTURE ONLY.

Plain-numpy float32 restatement of ``/root/reference/IPDnet2/IPDnet2.py`` (citations below are into that
file unless another is named).  Only ``tests/``, ``__graft_entry__.smoke()`` and bench's ``cpu_baseline``
leg may import it; nothing under ``fn-ssl_amd/`` does.

Parity status
-------------
* Everything that is pure torch in the reference is PINNED by ``tests/golden/g14_ipdnet2.npz``
  (``tests/golden/make_golden_ipdnet2.py`` runs the reference's own classes in the build container):
  ``LayerNorm`` (arch/base/norm.py:11-27), ``CausalConv1d`` with and without carried state (:45-82),
  ``SpatialNetLayer._fconv`` (:222-233), ``._full`` (:235-253), the two frequency poolings (:147-153), the time
  pooling (:345-349), ``FreqInverse`` (:23-43), decoder + output re-ordering (:357-364), and the whole
  ``SpatialNetLayer.forward`` / ``OnlineSpatialNet.forward`` ORCHESTRATION (:137-164, :331-368).
* The Mamba block is **parity unpinned**: ``mamba_ssm`` (state-spaces/mamba, version pinned nowhere in the
  reference, call sites :16-19,127,132,166-181) is not installed here and no checkpoint exists.  ``mamba()``
  below restates the published algorithm (Gu & Dao 2023, Alg. 2 + the package's ``Mamba.forward`` /
  ``selective_scan_ref`` semantics: in_proj -> causal depthwise conv + SiLU -> x_proj -> dt_proj + softplus ->
  h_t = exp(dt*A) h_{t-1} + dt*B_t*u_t, y_t = C_t.h_t + D*u_t -> * SiLU(z) -> out_proj) with the package's
  parameter names.  The whole-network fixtures were produced with a torch transcription of this same
  restatement plugged into the reference in place of the missing package, so they pin the orchestration
  around the block, not the block.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

_BF16 = False


def bf16_round(a) -> np.ndarray:
    """fp32 -> nearest-even bf16, returned as fp32 (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(a, dtype=F32).view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)).astype(np.uint32)
    return r.view(F32).reshape(np.shape(a))


class bf16_products:
    """``with bf16_products():`` restates FNSSL_PRECISION_BF16 (include/fnssl.h, fnssl_sn_encoder): BOTH operands of the
    encoder conv, the grouped frequency conv, the three products of the full-band branch (squeeze, Linear over F,
    unsqueeze) and the Mamba in / x / out projections are rounded to bf16 where they enter the product; accumulation,
    biases, activations, LayerNorm, the depthwise conv, dt_proj, the scan, FreqInverse and the decoder stay fp32.
    As on the device, the frequency conv below 16 bins and the full-band branch at other than 16 / 64 / 128 bins have
    no bf16 form and stay exact (only networks with fewer than 256 bins get there).  (In layer 0 the device applies the 5x time pooling to the operand of
    out_proj before it is rounded, this restatement after the product — the same rounding noise, not the same bits.)"""

    def __enter__(self):
        global _BF16
        self._old, _BF16 = _BF16, True
        return self

    def __exit__(self, *exc):
        global _BF16
        _BF16 = self._old
        return False


class _exact_if:
    """Switches the bf16 rounding off inside the block when ``cond`` (shapes without a bf16 kernel on the device)."""

    def __init__(self, cond):
        self.cond = cond

    def __enter__(self):
        global _BF16
        self._old = _BF16
        if self.cond:
            _BF16 = False

    def __exit__(self, *exc):
        global _BF16
        _BF16 = self._old
        return False


def _q(a):
    """An operand of a product that runs on bf16 MFMAs in FNSSL_PRECISION_BF16."""
    return bf16_round(a) if _BF16 else a


def _f(a):
    return np.asarray(a, dtype=F32)


def silu(x):
    x = _f(x)
    return (x / (F32(1) + np.exp(-x))).astype(F32)


def softplus(x):
    """torch.nn.functional.softplus (beta 1, threshold 20)."""
    x = _f(x)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, F32(20))))).astype(F32)


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim (arch/base/norm.py:11-27; seq_last only moves that dim)."""
    x = _f(x)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * _f(w) + _f(b)).astype(F32)


def causal_conv1d(x, w, b, state=None):
    """CausalConv1d.forward with look_ahead 0 (:66-76).  x [B, C, T], w [O, C, K], b [O];
    state [B, C, K-1] = the previous call's last K-1 input frames (None: zero left padding).
    Returns (y [B, O, T], new_state)."""
    x, w = _f(x), _f(w)
    K = w.shape[2]
    if state is None:
        xp = np.concatenate([np.zeros(x.shape[:2] + (K - 1,), F32), x], axis=-1)
    else:
        xp = np.concatenate([_f(state), x], axis=-1)
    T = x.shape[2]
    y = np.zeros((x.shape[0], w.shape[0], T), F32)
    for k in range(K):
        y += np.einsum("oc,bct->bot", _q(w[:, :, k]), _q(xp[:, :, k:k + T])).astype(F32)
    return (y + _f(b)[None, :, None]).astype(F32), xp[:, :, -(K - 1):].copy()


def grouped_conv_same(x, w, b, groups):
    """nn.Conv1d(C, C, K, groups, padding='same', zeros) along the last axis.  x [N, C, F], w [C, C/groups, K]."""
    x, w = _f(x), _f(w)
    N, C, F = x.shape
    cg, K = w.shape[1], w.shape[2]
    pad = (K - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (pad, K - 1 - pad)))
    y = np.zeros((N, C, F), F32)
    og = C // groups
    for g in range(groups):
        for k in range(K):
            y[:, g * og:(g + 1) * og] += np.einsum("oc,ncf->nof", _q(w[g * og:(g + 1) * og, :, k]),
                                                   _q(xp[:, g * cg:(g + 1) * cg, k:k + F])).astype(F32)
    return (y + _f(b)[None, :, None]).astype(F32)


def fconv(sd, p, x, groups=8):
    """SpatialNetLayer._fconv (:222-233) with ml = [LN(seq_last), Conv1d(H, H, k, groups, 'same'), PReLU(H)]
    (ctor :105-109 / :120-124).  x [B, F, T, H] -> same shape (WITHOUT the residual)."""
    x = _f(x)
    y = layer_norm(x, sd[p + ".0.weight"], sd[p + ".0.bias"])          # LN over H at every (b, f, t)
    B, F, T, H = y.shape
    y = y.transpose(0, 2, 3, 1).reshape(B * T, H, F)
    with _exact_if(F < 16):
        y = grouped_conv_same(y, sd[p + ".1.weight"], sd[p + ".1.bias"], groups)
    a = _f(sd[p + ".2.weight"])[None, :, None]
    y = np.where(y >= 0, y, a * y).astype(F32)                           # PReLU, one slope per channel
    return y.reshape(B, T, H, F).transpose(0, 3, 1, 2)


def full(sd, p, x):
    """SpatialNetLayer._full (:235-253), dropout off.  x [B, F, T, H] -> same (WITHOUT the residual)."""
    x = _f(x)
    y = layer_norm(x, sd[p + "norm_full.weight"], sd[p + "norm_full.bias"])
    ws, bs = _f(sd[p + "squeeze.0.weight"])[:, :, 0], _f(sd[p + "squeeze.0.bias"])
    with _exact_if(x.shape[1] not in (16, 64, 128)):
        s = silu(np.einsum("bfth,qh->bftq", _q(y), _q(ws)).astype(F32) + bs)        # [B, F, T, H']
        wf, bf = _f(sd[p + "full.weight"]), _f(sd[p + "full.bias"])
        s = (np.einsum("bftq,gf->bgtq", _q(s), _q(wf)).astype(F32) + bf[None, :, None, None]).astype(F32)   # Linear over F
        wu, bu = _f(sd[p + "unsqueeze.0.weight"])[:, :, 0], _f(sd[p + "unsqueeze.0.bias"])
        return silu(np.einsum("bftq,hq->bfth", _q(s), _q(wu)).astype(F32) + bu)


def avgpool_f(x, k):
    """fre_compress_* = AvgPool2d((1, k)) over F (:147-148,152-153; floor).  x [B, F, T, H]."""
    x = _f(x)
    B, F, T, H = x.shape
    return x[:, :F // k * k].reshape(B, F // k, k, T, H).mean(2, dtype=F32)


def avgpool_t(x, k):
    """time_pooling = AvgPool2d((k, 1)) over T of [B*F, T, H] (:345-349).  x [B, F, T, H]."""
    x = _f(x)
    B, F, T, H = x.shape
    return x[:, :, :T // k * k].reshape(B, F, T // k, k, H).mean(3, dtype=F32)


def mamba(sd, p, x, state=None):
    """One Mamba block (PARITY UNPINNED — see the module docstring).  x [S, T, D] -> ([S, T, D], state).

    sd[p + name], names of mamba_ssm.Mamba: in_proj.weight [2E, D], conv1d.weight [E, 1, K], conv1d.bias [E],
    x_proj.weight [R + 2N, E], dt_proj.weight [E, R], dt_proj.bias [E], A_log [E, N], D [E],
    out_proj.weight [D, E].  state = (conv_state [S, K-1, E], ssm_state [S, E, N]) or None."""
    x = _f(x)
    S, T, _ = x.shape
    w_in = _f(sd[p + "in_proj.weight"])
    E = w_in.shape[0] // 2
    xz = (_q(x) @ _q(w_in).T).astype(F32)
    xi, z = xz[..., :E], xz[..., E:]
    wc, bc = _f(sd[p + "conv1d.weight"])[:, 0, :], _f(sd[p + "conv1d.bias"])
    K = wc.shape[1]
    prev = np.zeros((S, K - 1, E), F32) if state is None else _f(state[0])
    xp = np.concatenate([prev, xi], axis=1)
    u = np.zeros((S, T, E), F32) + bc
    for k in range(K):
        u += xp[:, k:k + T, :] * wc[:, k]
    u = silu(u)
    wx = _f(sd[p + "x_proj.weight"])
    a_log = _f(sd[p + "A_log"])
    N = a_log.shape[1]
    R = wx.shape[0] - 2 * N
    dbl = (_q(u) @ _q(wx).T).astype(F32)
    dt = softplus((dbl[..., :R] @ _f(sd[p + "dt_proj.weight"]).T).astype(F32) + _f(sd[p + "dt_proj.bias"]))
    Bm, Cm = dbl[..., R:R + N], dbl[..., R + N:]
    A = -np.exp(a_log)
    h = np.zeros((S, E, N), F32) if state is None else _f(state[1]).copy()
    y = np.zeros((S, T, E), F32)
    Dp = _f(sd[p + "D"])
    for t in range(T):
        dA = np.exp(dt[:, t, :, None] * A[None])
        h = (dA * h + (dt[:, t, :, None] * Bm[:, t, None, :]) * u[:, t, :, None]).astype(F32)
        y[:, t] = (h * Cm[:, t, None, :]).sum(-1, dtype=F32) + Dp * u[:, t]
    y = (y * silu(z)).astype(F32)
    out = (_q(y) @ _q(_f(sd[p + "out_proj.weight"])).T).astype(F32)
    return out, (xp[:, -(K - 1):].copy(), h)


def mamba_step(sd, p, x_t, conv_state, ssm_state):
    """ONE frame of a Mamba block with explicit state — the recurrence the reference drives when ``inference=True``
    (IPDnet2.py:170-177: ``InferenceParams`` + ``mamba.forward(x[:, [i], :], inference_params)`` per frame, which in
    mamba_ssm dispatches to ``Mamba.step``).  PARITY UNPINNED like ``mamba``; restated from the published step form:
      conv_state <- roll(conv_state, -1); conv_state[..., -1] = x        (the last d_conv inputs of in_proj's x half)
      u  = SiLU(sum_k conv_state[..., k] * conv1d.weight[:, k] + conv1d.bias)
      dt = softplus(dt_proj(x_proj(u)[:R]) + dt_proj.bias);  B, C = x_proj(u)[R:R+N], [R+N:]
      ssm_state <- ssm_state * exp(dt A) + (dt B) u;   y = ssm_state . C + D u;   y <- y * SiLU(z);  out_proj(y)
    x_t [S, D]; conv_state [S, E, K] (oldest first; zeros before the first frame), ssm_state [S, E, N].
    Returns (out [S, D], conv_state, ssm_state) — new arrays."""
    x_t = _f(x_t)
    w_in = _f(sd[p + "in_proj.weight"])
    E = w_in.shape[0] // 2
    xz = (_q(x_t) @ _q(w_in).T).astype(F32)
    xi, z = xz[:, :E], xz[:, E:]
    wc, bc = _f(sd[p + "conv1d.weight"])[:, 0, :], _f(sd[p + "conv1d.bias"])
    conv_state = np.concatenate([_f(conv_state)[:, :, 1:], xi[:, :, None]], axis=2)
    u = np.zeros_like(xi) + bc
    for k in range(wc.shape[1]):
        u += conv_state[:, :, k] * wc[:, k]
    u = silu(u)
    wx = _f(sd[p + "x_proj.weight"])
    a_log = _f(sd[p + "A_log"])
    N = a_log.shape[1]
    R = wx.shape[0] - 2 * N
    dbl = (_q(u) @ _q(wx).T).astype(F32)
    dt = softplus((dbl[:, :R] @ _f(sd[p + "dt_proj.weight"]).T).astype(F32) + _f(sd[p + "dt_proj.bias"]))
    Bm, Cm = dbl[:, R:R + N], dbl[:, R + N:]
    A = -np.exp(a_log)
    dA = np.exp(dt[:, :, None] * A[None])
    ssm_state = (dA * _f(ssm_state) + (dt[:, :, None] * Bm[:, None, :]) * u[:, :, None]).astype(F32)
    y = (ssm_state * Cm[:, None, :]).sum(-1, dtype=F32) + _f(sd[p + "D"]) * u
    y = (y * silu(z)).astype(F32)
    out = (_q(y) @ _q(_f(sd[p + "out_proj.weight"])).T).astype(F32)
    return out, conv_state, ssm_state


def mamba_stepwise(sd, p, x):
    """The block driven frame by frame from zero state (IPDnet2.py:170-177).  x [S, T, D] -> [S, T, D]."""
    x = _f(x)
    S, T, _ = x.shape
    E = sd[p + "in_proj.weight"].shape[0] // 2
    K = sd[p + "conv1d.weight"].shape[2]
    N = sd[p + "A_log"].shape[1]
    cs, ss = np.zeros((S, E, K), F32), np.zeros((S, E, N), F32)
    outs = []
    for t in range(T):
        o, cs, ss = mamba_step(sd, p, x[:, t], cs, ss)
        outs.append(o)
    return np.stack(outs, axis=1)


def mamba_parallel_f64(sd, p, x):
    """Independent float64 evaluation of the same block in PARALLEL (non-recurrent) form, used only to cross-check
    ``mamba``: with c_t = sum_{r<=t} dt_r the recurrence h_t = exp(dt_t A) h_{t-1} + dt_t B_t u_t unrolls to
    h_t = sum_{s<=t} exp(A (c_t - c_s)) dt_s B_s u_s, evaluated directly (O(T^2)); the causal conv as an explicit
    Toeplitz sum.  x [S, T, D] -> [S, T, D] float64."""
    f8 = lambda k: np.asarray(sd[p + k], dtype=np.float64)   # noqa: E731
    x = np.asarray(x, dtype=np.float64)
    S, T, _ = x.shape
    w_in = f8("in_proj.weight")
    E = w_in.shape[0] // 2
    xz = x @ w_in.T
    xi, z = xz[..., :E], xz[..., E:]
    wc, bc = f8("conv1d.weight")[:, 0, :], f8("conv1d.bias")
    K = wc.shape[1]
    u = np.zeros((S, T, E)) + bc
    for t in range(T):
        for k in range(K):
            s_ = t - (K - 1) + k
            if s_ >= 0:
                u[:, t] += xi[:, s_] * wc[:, k]
    u = u / (1.0 + np.exp(-u))
    wx = f8("x_proj.weight")
    N = f8("A_log").shape[1]
    R = wx.shape[0] - 2 * N
    dbl = u @ wx.T
    pre = dbl[..., :R] @ f8("dt_proj.weight").T + f8("dt_proj.bias")
    dt = np.log1p(np.exp(-np.abs(pre))) + np.maximum(pre, 0.0)
    Bm, Cm = dbl[..., R:R + N], dbl[..., R + N:]
    A = -np.exp(f8("A_log"))                                     # [E, N]
    c = np.cumsum(dt, axis=1)                                     # [S, T, E]
    y = np.zeros((S, T, E))
    for t in range(T):
        decay = np.exp(A[None, None] * (c[:, t, None, :, None] - c[:, :t + 1, :, None]))     # [S, t+1, E, N]
        h = (decay * (dt[:, :t + 1, :, None] * Bm[:, :t + 1, None, :]) * u[:, :t + 1, :, None]).sum(axis=1)
        y[:, t] = (h * Cm[:, t, None, :]).sum(-1) + f8("D") * u[:, t]
    y = y * (z / (1.0 + np.exp(-z)))
    return y @ f8("out_proj.weight").T


def mamba_block(sd, p_norm, p_mamba, x, state=None, stepwise=False):
    """SpatialNetLayer._mamba (:166-181): LN, the block along T for every (b, f), WITHOUT the residual.
    ``stepwise`` = the reference's ``inference=True`` branch (:170-177): frame-by-frame recurrence from zero state."""
    B, F, T, H = x.shape
    y = layer_norm(x, sd[p_norm + ".weight"], sd[p_norm + ".bias"]).reshape(B * F, T, H)
    if stepwise:
        return mamba_stepwise(sd, p_mamba + ".", y).reshape(B, F, T, H), None
    y, st = mamba(sd, p_mamba + ".", y, state)
    return y.reshape(B, F, T, H), st


def layer_forward(sd, p, x, is_first, state=None, inference=False):
    """SpatialNetLayer.forward (:137-164).  x [B, F, T, H]; returns (x, state) with state = the two Mamba
    blocks' carried states.  ``inference``: the Mamba blocks are stepped frame by frame (:170-177)."""
    x = _f(x)
    x = x + fconv(sd, p + "fconv1", x)
    if is_first:
        x = avgpool_f(x, 2)
    x = x + full(sd, p, x)
    x = x + fconv(sd, p + "fconv2", x)
    if is_first:
        x = avgpool_f(x, 8)
    st = [None, None] if state is None else list(state)
    y, st[0] = mamba_block(sd, p + "norm_mhsa", p + "mhsa", x, st[0], stepwise=inference)
    x = x + y
    y, st[1] = mamba_block(sd, p + "norm_tconvffn", p + "tconvffn", x, st[1], stepwise=inference)
    x = x + y
    return x.astype(F32), st


def freq_inverse(sd, x, nfreq, ratio, out_dim):
    """FreqInverse.forward (:37-43).  x [B, H, T, Fc] -> [B, out_dim, T, nfreq] (after the tanh)."""
    x = _f(x)
    B, H, T, Fc = x.shape
    w, b = _f(sd["freq_inverse.trans2.weight"])[:, :, 0], _f(sd["freq_inverse.trans2.bias"])
    out = np.zeros((B, out_dim, nfreq, T), F32)
    for fi in range(nfreq // ratio):
        y = (np.einsum("oh,bht->bot", w, x[:, :, :, fi]).astype(F32) + b[None, :, None])   # [B, ratio*out, T]
        out[:, :, fi * ratio:(fi + 1) * ratio, :] += y.reshape(B, out_dim, -1, T)
    return np.tanh(out.transpose(0, 1, 3, 2)).astype(F32)


def num_layers_of(sd):
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))


def forward(sd, x, time_compression_ratio=5, fre_compression_ratio=16, state=None, inference=False):
    """OnlineSpatialNet.forward (:331-368) with time_compression_layer = 0 (``inference`` as the reference's flag:
    per-frame Mamba stepping, :170-177; not combinable with ``state``).
    x [B, dim_input, F, T] -> [B, T // ratio, 2F, dim_output // 4, 2]  (and the carried state when asked:
    pass state={} for the first chunk; T must then be a multiple of the time ratio)."""
    x = _f(x).transpose(0, 2, 3, 1)                                     # [B, F, T, H0]   (:333)
    B, F, T, H0 = x.shape
    enc_state = None if not state else state.get("enc")
    y, enc_new = causal_conv1d(x.reshape(B * F, T, H0).transpose(0, 2, 1), sd["encoder.weight"], sd["encoder.bias"],
                               enc_state)
    x = y.transpose(0, 2, 1).reshape(B, F, T, -1)                        # (:335)
    new_state = {"enc": enc_new}
    for l in range(num_layers_of(sd)):
        x, st = layer_forward(sd, "layers.%d." % l, x, l == 0, None if not state else state.get("l%d" % l), inference)
        new_state["l%d" % l] = st
        if l == 0:
            x = avgpool_t(x, time_compression_ratio)                    # (:343-349)
    B, Fc, T2, H = x.shape
    do = sd["decoder.weight"].shape[0]
    y = freq_inverse(sd, x.transpose(0, 3, 2, 1), F, fre_compression_ratio, do)     # [B, do, T2, F]  (:357-358)
    y = y.transpose(0, 3, 2, 1)                                          # [B, F, T2, do]  (:359)
    y = (y @ _f(sd["decoder.weight"]).T + _f(sd["decoder.bias"])).astype(F32)      # (:360)
    y = y.transpose(0, 2, 1, 3).reshape(B, T2, F, 2, -1).transpose(0, 1, 3, 2, 4)   # (:363)
    y = np.ascontiguousarray(y).reshape(B, T2, 2, F * 2, -1).transpose(0, 1, 3, 4, 2)   # (:364)
    y = np.ascontiguousarray(y)
    return (y, new_state) if state is not None else y


def flops_per_frame(dim_input=10, dim_output=16, num_layers=8, H=96, Hs=8, F=256, ke=5, kf=5, groups=8, N=16,
                    Kc=4, ratio_f=16, ratio_t=5):
    """Algorithmic flop (2 per MAC of every matmul / conv; norms, activations and the scan's element-wise
    part counted at face value) per INPUT frame of one utterance."""
    E, R = 2 * H, -(-H // 16)
    fc = lambda f: f * 2 * H * (H // groups) * kf                                              # noqa: E731
    fl = lambda f: f * (2 * H * Hs * 2) + 2 * Hs * f * f                                         # noqa: E731
    mb = lambda f: f * (2 * H * 2 * E + 2 * E * Kc + 2 * E * (R + 2 * N) + 2 * R * E + 7 * E * N + 2 * E * H)   # noqa: E731
    total = F * 2 * dim_input * ke * H                       # encoder
    total += fc(F) + fl(F // 2) + fc(F // 2) + 2 * mb(F // ratio_f)     # layer 0
    per = fc(F // ratio_f) * 2 + fl(F // ratio_f) + 2 * mb(F // ratio_f)
    total += (num_layers - 1) * per / ratio_t
    total += (F // ratio_f) * 2 * H * ratio_f * dim_output / ratio_t + F * 2 * dim_output * dim_output / ratio_t
    return float(total)
