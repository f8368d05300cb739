"""CPU oracle for the FN-SSL DP-IPD forward path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference algorithm.  It exists so
that the HIP path can be checked on a GPU box where ``/root/reference`` does not
exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; nothing under ``fn-ssl_amd/`` does, and the
product path has no CPU fallback.

Parity is PINNED: ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference/FN-SSL/Lightning/{Model,Module,utils_}.py``) in the build
container, runs it on seeded inputs with weights from
``fnssl.weights`` and commits the outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against them.

Each function cites the reference lines it restates (paths relative to
``/root/reference/``).  All arithmetic is float32 unless noted.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
WIN_LEN = 512
HOP = 256
NFFT = 512
NBIN = 257
SEG_FRAMES = 12


# --------------------------------------------------------------------------- #
# front end
# --------------------------------------------------------------------------- #
def hann_periodic(n: int = WIN_LEN) -> np.ndarray:
    """torch.hann_window(n) (periodic).  FN-SSL/Module.py:61."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(F32)


def n_frames(ns: int, win_len: int = WIN_LEN, hop: int = HOP) -> int:
    """FN-SSL/Module.py:56: floor((ns - win_len) / hop + 1)."""
    return int(np.floor((ns - win_len) / hop + 1))


def stft(signal: np.ndarray, hop: int = HOP, center: bool = False) -> np.ndarray:
    """STFT.forward, FN-SSL/Module.py:48-68.

    signal [nb, ns, nch] float32 -> complex64 [nb, 257, nt, nch]; Hann-512
    periodic window, hop 256, center=False, not normalised.  The DFT itself is
    evaluated in float64 and rounded once, i.e. it is the exact value the
    reference's float32 FFT approximates.

    ``hop=320, center=True`` restates IPDnet2's transform (IPDnet2/Module.py:47-64): torch.stft's centred
    framing = the signal extended by 256 reflected samples at both ends (pad_mode 'reflect', edge sample not
    repeated), nt = floor(ns / hop + 1) (:55).
    """
    signal = np.asarray(signal, dtype=F32)
    nb, ns, nch = signal.shape
    if center:
        signal = np.pad(signal, ((0, 0), (WIN_LEN // 2, WIN_LEN // 2), (0, 0)), mode="reflect")
        nt = int(np.floor(ns / hop + 1))
    else:
        nt = n_frames(ns, hop=hop)
    win = hann_periodic()
    idx = (np.arange(nt)[:, None] * hop + np.arange(WIN_LEN)[None, :])  # [nt, 512]
    out = np.zeros((nb, NBIN, nt, nch), dtype=np.complex64)
    for c in range(nch):
        frames = signal[:, :, c][:, idx] * win[None, None, :]            # f32 product
        spec = np.fft.rfft(frames.astype(np.float64), n=NFFT, axis=-1)   # [nb, nt, 257]
        out[:, :, :, c] = np.transpose(spec, (0, 2, 1)).astype(np.complex64)
    return out


def n_pairs(nch: int, ch_mode: str) -> int:
    return nch * (nch - 1) // 2 if ch_mode == "MM" else nch - 1


def pair_list(nch: int, ch_mode: str):
    """Mic index pairs in the row order AddChToBatch produces.

    FN-SSL/Module.py:387-393 ('M': (0, j)), :397-402 ('MM': i<j, i-major).
    """
    if ch_mode == "M":
        return [(0, j) for j in range(1, nch)]
    if ch_mode == "MM":
        return [(i, j) for i in range(nch - 1) for j in range(i + 1, nch)]
    raise ValueError("ch_mode must be 'M' or 'MM'")


def add_ch_to_batch(data: np.ndarray, ch_mode: str) -> np.ndarray:
    """AddChToBatch.forward, FN-SSL/Module.py:383-404.

    [nb, nch, nf, nt] complex -> [nb*np, 2, nf, nt] complex64.
    """
    nb, nch = data.shape[:2]
    pairs = pair_list(nch, ch_mode)
    out = np.zeros((nb * len(pairs), 2) + data.shape[2:], dtype=np.complex64)
    for b in range(nb):
        for p, (i, j) in enumerate(pairs):
            out[b * len(pairs) + p, 0] = data[b, i]
            out[b * len(pairs) + p, 1] = data[b, j]
    return out


def remove_ch_from_batch(data: np.ndarray, nb: int) -> np.ndarray:
    """RemoveChFromBatch.forward, FN-SSL/Module.py:412-421."""
    nmic = data.shape[0] // nb
    return np.ascontiguousarray(data.reshape((nb, nmic) + data.shape[1:])).astype(F32)


def forgetting_coefs(nt: int, sample_length: int = 298):
    """Per-frame (a_t, b_t) of the recursion mu_t = a_t*mu_{t-1} + b_t*mean_t.

    FN-SSL/utils.py:26-44.  For t < sample_length the reference builds
    ``alp`` as a float32 tensor (``torch.min(torch.tensor([...]))``) and forms
    ``1 - alp`` in float32; afterwards it uses the Python double ``alpha`` whose
    two coefficients are each rounded to float32 when they meet the tensor.
    t = 0 gives alp = -1, i.e. mu_0 = 2*mean_0.
    """
    alpha = (sample_length - 1) / (sample_length + 1)
    a = np.empty(nt, dtype=F32)
    b = np.empty(nt, dtype=F32)
    for t in range(nt):
        if t < sample_length:
            alp = F32(min((t - 1) / (t + 1), alpha))
            a[t] = alp
            b[t] = F32(1.0) - alp
        else:
            a[t] = F32(alpha)
            b[t] = F32(1.0 - alpha)
    return a, b


def forgetting_norm(mag: np.ndarray, sample_length: int = 298) -> np.ndarray:
    """forgetting_norm, FN-SSL/utils.py:9-55.  mag [B,C,F,T] -> mu [B,1,1,T]."""
    assert mag.ndim == 4
    B, C, Fq, T = mag.shape
    m = mag.reshape(B, C * Fq, T).astype(F32)
    a, b = forgetting_coefs(T, sample_length)
    mu = np.zeros((B,), dtype=F32)
    out = np.empty((B, T), dtype=F32)
    for t in range(T):
        mean_t = m[:, :, t].mean(axis=1, dtype=F32)
        mu = (a[t] * mu).astype(F32) + (b[t] * mean_t).astype(F32)
        out[:, t] = mu
    return out.reshape(B, 1, 1, T)


def data_preprocess(mic_sig: np.ndarray, ch_mode: str = "MM", eps: float = 1e-6,
                    sample_length: int = 298) -> np.ndarray:
    """Input half of data_preprocess, FN-SSL/Lightning/main.py:200-225
    (= FN-SSL/Learner.py:392-414).

    mic_sig [nb, ns, nch] -> x [nb*np, 4, 256, nt] float32 with channels
    [Re i, Re j, Im i, Im j], normalised by (mu_t + eps), DC bin dropped
    (fre_range_used = 1..256, main.py:130).
    """
    spec = stft(mic_sig)                               # [nb, 257, nt, nch]
    spec = np.transpose(spec, (0, 3, 1, 2))            # [nb, nch, 257, nt]  main.py:207
    reb = add_ch_to_batch(spec, ch_mode)               # [nb', 2, 257, nt]
    mag = np.abs(reb).astype(F32)
    mu = forgetting_norm(mag, sample_length)           # [nb', 1, 1, nt]
    den = (mu + F32(eps)).astype(F32)
    re = (reb.real.astype(F32) / den).astype(F32)
    im = (reb.imag.astype(F32) / den).astype(F32)
    x = np.concatenate([re, im], axis=1)               # [nb', 4, 257, nt]
    return np.ascontiguousarray(x[:, :, 1:NBIN, :])


def array_preprocess(mic_sig: np.ndarray, eps: float = 1e-6, sample_length: int = 280, hop: int = HOP,
                     center: bool = False) -> np.ndarray:
    """Input half of IPDnet's data_preprocess, IPDnet/runIPDnetOn.py:240-254: all channels are
    normalised together.  mic_sig [nb, ns, nch] -> x [nb, 2*nch, 256, nt], channels [Re all, Im all].
    With ``sample_length=249, hop=320, center=True`` it is IPDnet2's (IPDnet2/run_IPDnet2.py:277-288)."""
    spec = np.transpose(stft(mic_sig, hop, center), (0, 3, 1, 2))   # [nb, nch, 257, nt]   :246
    mu = forgetting_norm(np.abs(spec).astype(F32), sample_length)           # :248-249
    den = (mu + F32(eps)).astype(F32)
    x = np.concatenate([(spec.real.astype(F32) / den).astype(F32), (spec.imag.astype(F32) / den).astype(F32)], axis=1)
    return np.ascontiguousarray(x[:, :, 1:NBIN, :])                          # fre_range_used :125


# --------------------------------------------------------------------------- #
# network
# --------------------------------------------------------------------------- #
def _sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def bf16_round(a) -> np.ndarray:
    """fp32 -> nearest-even bf16, returned as fp32 (what v_cvt_pk_bf16_f32 / the bf16 weight packer do)."""
    u = np.ascontiguousarray(a, dtype=F32).view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)).astype(np.uint32)
    return r.view(F32).reshape(np.shape(a))


def lstm_dir(x: np.ndarray, w_ih, w_hh, b_ih, b_hh, reverse: bool = False,
             h0=None, c0=None, bf16: bool = False) -> np.ndarray:
    """One direction of a 1-layer batch_first nn.LSTM with zero initial state.

    PyTorch semantics (gate order i, f, g, o):
        g_t = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh
        c_t = sigmoid(f) * c_{t-1} + sigmoid(i) * tanh(g)
        h_t = sigmoid(o) * tanh(c_t)
    x [N, T, I] -> h [N, T, H].  Used by FN-SSL/Model.py:25-29,38,46.

    bf16=True restates the bf16-MFMA kernels (csrc/lstm_bf16.h): W_ih, W_hh and the operands x_t, h_{t-1} are
    rounded to bf16 where they enter the product; bias, accumulation, gates, cell state and the stored h are fp32.
    """
    x = np.asarray(x, dtype=F32)
    N, T, _ = x.shape
    H = w_hh.shape[1]
    if bf16:
        x, w_ih, w_hh = bf16_round(x), bf16_round(w_ih), bf16_round(w_hh)
    xw = (x.reshape(N * T, -1) @ w_ih.T.astype(F32)).reshape(N, T, 4 * H)
    xw = (xw + (b_ih + b_hh).astype(F32)).astype(F32)
    whhT = np.ascontiguousarray(w_hh.T.astype(F32))
    h = np.zeros((N, H), dtype=F32) if h0 is None else h0.astype(F32)
    c = np.zeros((N, H), dtype=F32) if c0 is None else c0.astype(F32)
    out = np.empty((N, T, H), dtype=F32)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = (xw[:, t] + (bf16_round(h) if bf16 else h) @ whhT).astype(F32)
        i = _sigmoid(g[:, 0:H])
        f = _sigmoid(g[:, H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H], dtype=F32)
        o = _sigmoid(g[:, 3 * H:4 * H])
        c = (f * c + i * gg).astype(F32)
        h = (o * np.tanh(c, dtype=F32)).astype(F32)
        out[:, t] = h
    return out


def lstm(x: np.ndarray, sd: dict, prefix: str, bidirectional: bool, bf16: bool = False) -> np.ndarray:
    """nn.LSTM forward (output only); bi-dir output is [fwd || bwd]."""
    p = lambda n: sd[prefix + n]  # noqa: E731
    fwd = lstm_dir(x, p("weight_ih_l0"), p("weight_hh_l0"), p("bias_ih_l0"), p("bias_hh_l0"), bf16=bf16)
    if not bidirectional:
        return fwd
    bwd = lstm_dir(x, p("weight_ih_l0_reverse"), p("weight_hh_l0_reverse"),
                   p("bias_ih_l0_reverse"), p("bias_hh_l0_reverse"), reverse=True, bf16=bf16)
    return np.concatenate([fwd, bwd], axis=-1)


def fnblock_forward(sd: dict, prefix: str, x: np.ndarray, fb_skip=None, *,
                    is_first: bool, is_online: bool, bf16: bool = False):
    """FNblock.forward in eval mode, FN-SSL/Model.py:31-50.

    x [nb, nt, nf, C] -> (x_out [nb, nt, nf, Hn], fb_skip [nb*nt, nf, 2*Hf],
    nb_skip [nb*nf, nt, Hn]).  The incoming ``nb_skip`` argument is overwritten
    by the reference (:34) so it is not a parameter here; dropout is identity.
    """
    nb, nt, nf, _ = x.shape
    nb_skip = np.transpose(x, (0, 2, 1, 3)).reshape(nb * nf, nt, -1)          # :34
    xf = x.reshape(nb * nt, nf, -1)                                           # :35
    if not is_first:
        xf = (xf + fb_skip).astype(F32)                                       # :36-37
    f = lstm(xf, sd, prefix + "fullLstm.", True, bf16)                        # :38
    fb_out = f                                                                # :39
    v = np.transpose(f.reshape(nb, nt, nf, -1), (0, 2, 1, 3)).reshape(nb * nf, nt, -1)  # :41
    if is_first:
        v = np.concatenate([v, nb_skip], axis=-1)                             # :42-43
    else:
        v = (v + nb_skip).astype(F32)                                         # :44-45
    n = lstm(v, sd, prefix + "narrLstm.", not is_online, bf16)                # :46
    nb_out = n                                                                # :47
    xo = np.transpose(n.reshape(nb, nf, nt, -1), (0, 2, 1, 3))                # :49
    return np.ascontiguousarray(xo), fb_out, nb_out


def head_forward(sd: dict, x: np.ndarray) -> np.ndarray:
    """Pool + emb2ipd + tanh + re|im packing, FN-SSL/Model.py:79-87.

    x [nb, nt, nf, 256] -> [nb, nt//12, 2*nf].  AvgPool2d((12,1)) floors.
    """
    nb, nt, nf, C = x.shape
    nt2 = nt // SEG_FRAMES
    xs = np.transpose(x, (0, 2, 1, 3))[:, :, :nt2 * SEG_FRAMES]               # [nb, nf, nt2*12, C]
    pooled = (xs.reshape(nb, nf, nt2, SEG_FRAMES, C).sum(axis=3, dtype=F32) / F32(SEG_FRAMES)).astype(F32)
    ipd = np.tanh((pooled @ sd["emb2ipd.weight"].T.astype(F32) + sd["emb2ipd.bias"]).astype(F32),
                  dtype=F32)                                                  # [nb, nf, nt2, 2]
    ipd = np.transpose(ipd, (0, 2, 1, 3))                                     # [nb, nt2, nf, 2]
    return np.ascontiguousarray(np.concatenate([ipd[..., 0], ipd[..., 1]], axis=2))


def fnssl_forward(sd: dict, x: np.ndarray, is_online: bool = True, is_doa: bool = False,
                  bf16: bool = False) -> np.ndarray:
    """FN_SSL.forward, FN-SSL/Model.py:72-90.  x [nb', 4, nf, nt] -> [nb', nt//12, 2nf].
    bf16=True: the module after .bfloat16() — bf16 parameters and input, LSTM products on bf16 operands
    (fp32 accumulate, residual adds, head and tensors in fp32)."""
    if bf16:
        sd = {k: bf16_round(v) for k, v in sd.items()}
        x = bf16_round(x)
    x = np.transpose(np.asarray(x, dtype=F32), (0, 3, 2, 1))                  # :73
    x, fb, _ = fnblock_forward(sd, "block_1.", x, is_first=True, is_online=is_online, bf16=bf16)
    x, fb, _ = fnblock_forward(sd, "block_2.", x, fb, is_first=False, is_online=is_online, bf16=bf16)
    x, fb, _ = fnblock_forward(sd, "block_3.", x, fb, is_first=False, is_online=is_online, bf16=bf16)
    res = head_forward(sd, x)
    if is_doa:
        res = (res @ sd["ipd2doa.weight"].T.astype(F32) + sd["ipd2doa.bias"]).astype(F32)  # :88-89
    return res


def predict_step(sd: dict, batch: np.ndarray, ch_mode: str = "MM", is_online: bool = True) -> np.ndarray:
    """MyModel.predict_step, FN-SSL/Lightning/main.py:184-189.  batch [nb, nch, ns]."""
    x = data_preprocess(np.transpose(batch, (0, 2, 1)), ch_mode)
    return fnssl_forward(sd, x, is_online=is_online)


# --------------------------------------------------------------------------- #
# work accounting (SURVEY.md §8d / BASELINE.md §3)
# --------------------------------------------------------------------------- #
def flops_per_tf_point(is_online: bool = True, input_size: int = 4) -> int:
    """LSTM matmul flops (2 per MAC) for one (pair, bin, frame) TF point."""
    def l(i, h, ndir):
        return 2 * 4 * h * (i + h) * ndir
    hn, nd = (256, 1) if is_online else (128, 2)
    total = l(input_size, 128, 2) + l(256 + input_size, hn, nd)
    total += 2 * (l(256, 128, 2) + l(256, hn, nd))
    return total


# --------------------------------------------------------------------------- #
# IPD -> DOA back end (next row after the forward path)
# --------------------------------------------------------------------------- #
def dpipd_templates(mic_location: np.ndarray, nele: int = 37, nazi: int = 73, nf: int = 257,
                    fre_max: float = 8000.0, ch_mode: str = "MM", speed: float = 340.0):
    """DPIPD.__init__ + data_adjust, FN-SSL/Lightning/Module.py:429-463,505-519.

    Returns (template complex64 [nele, nazi, nf, np], [ele_candidate, azi_candidate]).
    """
    mic_location = np.asarray(mic_location, dtype=np.float64)
    nmic = mic_location.shape[-2]
    ele = np.linspace(0, np.pi, nele)
    azi = np.linspace(-np.pi, np.pi, nazi)
    fre = np.linspace(0.0, fre_max, nf)
    r = np.stack([np.outer(np.sin(ele), np.cos(azi)), np.outer(np.sin(ele), np.sin(azi)),
                  np.tile(np.cos(ele), [nazi, 1]).transpose()], axis=2)            # [nele, nazi, 3]
    ipd = np.empty((nele, nazi, nf, nmic, nmic))
    for m1 in range(nmic):
        for m2 in range(nmic):
            itd = np.dot(r, mic_location[m2, :] - mic_location[m1, :]) / speed     # :450
            ipd[:, :, :, m1, m2] = -2 * np.pi * fre[None, None, :] * itd[:, :, None]
    tmpl = np.exp(1j * ipd)
    pairs = pair_list(nmic, ch_mode)
    out = np.empty((nele, nazi, nf, len(pairs)), dtype=np.complex64)
    for p, (i, j) in enumerate(pairs):
        out[..., p] = tmpl[..., i, j]
    return out, [ele, azi]


def dpipd_of_sources(source_doa: np.ndarray, mic_location, nf: int = 257, fre_max: float = 8000.0, ch_mode: str = "MM",
                     speed: float = 340.0) -> np.ndarray:
    """DPIPD.forward(source_doa), FN-SSL/Lightning/Module.py:464-498: source_doa [nb, ntime, 2, nsource] ->
    complex64 [nb, ntime, nf, np, nsource].  Note the sign: ITD[m1, m2] = r . (mic[m1] - mic[m2]) / c (:488) and the
    phase is (-2 pi f ITD) * (-1) (:489-490) — the conjugate convention of the template bank of __init__."""
    doa = np.asarray(source_doa).transpose(0, 1, 3, 2)                        # (nb, ntime, nsource, 2)  :472
    mic = np.asarray(mic_location, dtype=np.float64)
    nmic = mic.shape[-2]
    nb, ntime, nsource = doa.shape[:3]
    fre = np.linspace(0.0, fre_max, nf)
    r = np.stack([np.sin(doa[..., 0]) * np.cos(doa[..., 1]), np.sin(doa[..., 0]) * np.sin(doa[..., 1]), np.cos(doa[..., 0])],
                 axis=3)                                                      # :484-486
    ipd = np.empty((nb, ntime, nsource, nf, nmic, nmic))
    for m1 in range(nmic):
        for m2 in range(nmic):
            itd = np.dot(r, mic[m1, :] - mic[m2, :]) / speed                  # :488
            ipd[:, :, :, :, m1, m2] = -2 * np.pi * fre[None, None, None, :] * itd[..., None] * (-1)   # :489-490
    full = np.exp(1j * ipd)
    pairs = pair_list(nmic, ch_mode)                                          # data_adjust :500-514
    out = np.empty((nb, ntime, nsource, nf, len(pairs)), dtype=np.complex64)
    for p, (i, j) in enumerate(pairs):
        out[..., p] = full[..., i, j]
    return out.transpose(0, 1, 3, 4, 2)                                       # (nb, ntime, nf, np, nsource)  :495


def dpipd_targets(doa: np.ndarray, vad: np.ndarray, mic_location, ch_mode: str = "MM", use_vad: bool = True,
                  nfft: int = 512, fs: int = 16000, speed: float = 340.0):
    """Ground-truth half of MyModel.data_preprocess, FN-SSL/Lightning/main.py:227-262: doa [nb, nseg, 2, ns], vad [nb, nseg,
    nvad, ns] -> (gt_batch['ipd'] [nb, nseg, 2 * 256, np] float32, gt_batch['vad_sources'] [nb, nseg, ns])."""
    dp = dpipd_of_sources(doa, mic_location, nfft // 2 + 1, fs / 2, ch_mode, speed)
    used = slice(1, nfft // 2 + 1)                                            # fre_range_used, main.py:130
    ipd = np.concatenate((dp.real[:, :, used], dp.imag[:, :, used]), axis=2).astype(np.float32)     # :237-238
    vmean = np.asarray(vad, dtype=np.float32).mean(axis=2)                    # :243
    if use_vad:                                                               # :249-257, th = 0
        gate = (vmean > 0).astype(np.float32)
        ipd = ipd * gate[:, :, None, None, :]
    return ipd.sum(axis=-1, dtype=np.float32), vmean                          # :258


def template_bank(template: np.ndarray, doa_candidate):
    """The candidate bank PredDOA.predgt2DOA actually searches, Module.py:702-716: real|imag of
    bins 1..256 concatenated along frequency, elevation fixed to the middle row, azimuth the upper
    half of the grid; candidates replaced by ele = pi/2, azi = linspace(0, pi, 37)."""
    nele, nazi = template.shape[:2]
    t = np.concatenate((template.real[:, :, 1:NBIN, :], template.imag[:, :, 1:NBIN, :]), axis=2).astype(F32)
    t = t[int((nele - 1) / 2):int((nele - 1) / 2) + 1, int((nazi - 1) / 2):nazi, :, :]
    cand = [np.linspace(np.pi / 2, np.pi / 2, 1), np.linspace(0, np.pi, 37)]
    return np.ascontiguousarray(t), cand


def source_detect_localize(pred_ipd: np.ndarray, bank: np.ndarray, cand, max_num_sources: int = 1,
                           source_num_mode: str = "kNum"):
    """SourceDetectLocalize.forward, meth_mode 'IDL', Module.py:525-577.

    pred_ipd [nb, nt, 2nf, np] (= RemoveChFromBatch(pred).permute(0, 2, 3, 1)), bank [nele, nazi, 2nf, np]
    -> (doa [nb, nt, 2, ns], vad [nb, nt, ns], ss [nb, nt, nele, nazi])
    """
    pred = np.array(pred_ipd, dtype=F32, copy=True)
    nb, nt, nf2, npair = pred.shape
    nele, nazi = bank.shape[:2]
    flat = bank.reshape(nele * nazi, nf2 * npair).astype(F32)                      # [ncand, 2nf*np]
    norm = F32(npair * nf2 / 2)
    doas = np.zeros((nb, nt, 2, max_num_sources), dtype=F32)
    vads = np.zeros((nb, nt, max_num_sources), dtype=F32)
    ss0 = None
    for s in range(max_num_sources):
        m = (pred.reshape(nb, nt, -1) @ flat.T / norm).astype(F32)                 # [nb, nt, ncand]
        if ss0 is None:
            ss0 = m.reshape(nb, nt, nele, nazi).copy()
        idx = m.argmax(axis=2)
        ei, ai = np.unravel_index(idx, (nele, nazi))
        doas[:, :, 0, s] = cand[0][ei]
        doas[:, :, 1, s] = cand[1][ai]
        tm = flat[idx].reshape(nb, nt, nf2, npair)                                 # chosen templates
        ratio = (tm * pred).sum(axis=(2, 3), dtype=F32) / (tm * tm).sum(axis=(2, 3), dtype=F32)
        vads[:, :, s] = 1 if source_num_mode == "kNum" else ratio
        pred = (pred - ratio[:, :, None, None] * tm).astype(F32)
    return doas, vads, ss0


def source_detect_localize_pd(pred_ipd: np.ndarray, bank: np.ndarray, cand, max_num_sources: int = 1,
                              source_num_mode: str = "kNum"):
    """SourceDetectLocalize.forward, meth_mode 'PD' (peak detection), Module.py:580-622.

    The spatial spectrum of the first pass; the last azimuth column is dropped as redundant (:581); a cell is a peak when it
    is strictly larger than its 8 neighbours, azimuth circular over the remaining columns, elevation CLAMPED (:583-598 — the
    clamp compares rows 0 and nele - 1 with themselves, so they never hold a peak); per frame the peaks are sorted by value,
    descending, ties in ascending flat-index order (python's stable `sorted`, :608-609), and the first max_num_sources kept.
    What the reference's slice assignment `pred_DOAs[b, t, :, :] = pred_DOA.transpose(1, 0)` (:615) then does, measured on the
    real reference (tests/golden/make_golden_pd.py): the picked indices are a LIST of one-element tensors, so pred_DOA is
    [n, 1, 2] and its transpose [1, n, 2] — for max_num_sources = 2 that lands as doa[b, t, SOURCE, (ele, azi)], the
    TRANSPOSE of the 'IDL' branch's [.., (ele, azi), source] layout (reproduced here as it is: a drop-in returns what the
    reference returns); for any other number of sources the assignment raises, as does a frame without a peak; a frame with
    exactly one peak broadcasts it to both sources.  Same here (ValueError where the reference raises RuntimeError).
    -> (doa [nb, nt, 2 (source), 2 (ele, azi)], vad [nb, nt, 2], ss [nb, nt, nele, nazi])"""
    pred = np.asarray(pred_ipd, dtype=F32)
    nb, nt, nf2, npair = pred.shape
    nele, nazi = bank.shape[:2]
    flat = bank.reshape(nele * nazi, nf2 * npair).astype(F32)
    ss = (pred.reshape(nb, nt, -1) @ flat.T / F32(npair * nf2 / 2)).astype(F32).reshape(nb, nt, nele, nazi)
    ns = max_num_sources
    if ns != 2:
        raise ValueError("the reference's 'PD' branch only runs with max_num_sources = 2 (Module.py:615 raises otherwise)")
    doas = np.zeros((nb, nt, 2, ns), dtype=F32)
    vads = np.zeros((nb, nt, ns), dtype=F32)
    w = nazi - 1
    for b in range(nb):
        for t in range(nt):
            g = ss[b, t, :, :w]
            found = []
            for e in range(nele):
                for a in range(w):
                    v, ok = g[e, a], True
                    for de in (-1, 0, 1):
                        for da in (-1, 0, 1):
                            if de == 0 and da == 0:
                                continue
                            if not v > g[min(max(e + de, 0), nele - 1), (a + da) % w]:
                                ok = False
                    if ok:
                        found.append((e * nazi + a, v))
            found.sort(key=lambda kv: -kv[1])                                   # stable: ties stay in index order
            found = found[:ns]
            if len(found) == 0:
                raise ValueError("frame (%d, %d): no peak" % (b, t))
            for s in range(ns):
                k, v = found[s if len(found) > 1 else 0]
                doas[b, t, s, 0] = cand[0][k // nazi]                            # [source, (ele, azi)]: see above
                doas[b, t, s, 1] = cand[1][k % nazi]
                vads[b, t, s] = 1 if source_num_mode == "kNum" else v
    return doas, vads, ss


def pred_to_doa(pred: np.ndarray, nb: int, mic_location, ch_mode: str = "MM", max_num_sources: int = 1,
                source_num_mode: str = "kNum", speed: float = 340.0):
    """PredDOA.predgt2DOA (prediction half), Module.py:690-727: network output
    [nb*np, nt, 2nf] -> dict(doa, vad_sources, spatial_spectrum)."""
    tmpl, cand = dpipd_templates(mic_location, 37, 73, NBIN, 8000.0, ch_mode, speed)
    bank, cand = template_bank(tmpl, cand)
    rebatch = np.transpose(remove_ch_from_batch(pred, nb), (0, 2, 3, 1))           # [nb, nt, 2nf, np]
    doa, vad, ss = source_detect_localize(rebatch, bank, cand, max_num_sources, source_num_mode)
    return {"doa": doa, "vad_sources": vad, "spatial_spectrum": ss}


# --------------------------------------------------------------------------- #
# IPDnet, fixed array.  Reference IPDnet/FixedAarryIPDnet.py
# --------------------------------------------------------------------------- #
def ipdnet_block(sd, prefix, x, skip, is_online, bf16=False):
    """IPDnet FNblock.forward (eval), FixedAarryIPDnet.py:29-40.  x [nb, nt, nf, C], skip [nb, nt, nf, Cs]:
    full-band BiLSTM, concat skip, narrow-band LSTM, concat skip -> [nb, nt, nf, Hn + Cs]."""
    nb, nt, nf, _ = x.shape
    f = lstm(x.reshape(nb * nt, nf, -1), sd, prefix + "fullLstm.", True, bf16)                   # :31-32
    v = np.concatenate([f, skip.reshape(nb * nt, nf, -1)], axis=-1)                              # :34
    v = np.transpose(v.reshape(nb, nt, nf, -1), (0, 2, 1, 3)).reshape(nb * nf, nt, -1)           # :35
    n = lstm(v, sd, prefix + "narrLstm.", not is_online, bf16)                                   # :36
    n = np.concatenate([n, np.transpose(skip, (0, 2, 1, 3)).reshape(nb * nf, nt, -1)], axis=-1)  # :38
    return np.ascontiguousarray(np.transpose(n.reshape(nb, nf, nt, -1), (0, 2, 1, 3)))           # :39


def conv3x3_pad12(x, w):
    """nn.Conv2d(k=3x3, stride 1, padding (1, 2), bias=False) on x [nb, Cin, F, T] -> [nb, Cout, F, T + 2]."""
    nb, cin, F, T = x.shape
    xp = np.zeros((nb, cin, F + 2, T + 4), dtype=F32)
    xp[:, :, 1:F + 1, 2:T + 2] = x
    out = np.zeros((nb, w.shape[0], F, T + 2), dtype=F32)
    for df in range(3):
        for dt in range(3):
            patch = xp[:, :, df:df + F, dt:dt + T + 2]                     # [nb, cin, F, T+2]
            out += np.einsum("oc,bcft->boft", w[:, :, df, dt].astype(F32), patch, optimize=True).astype(F32)
    return out


def avgpool_t(x, k):
    """nn.AvgPool2d((1, k)) on [nb, C, F, T] (floors)."""
    nb, c, F, T = x.shape
    t2 = T // k
    return (x[:, :, :, :t2 * k].reshape(nb, c, F, t2, k).sum(axis=4, dtype=F32) / F32(k)).astype(F32)


def caus_cnn_block(sd, prefix, x, bf16=False):
    """CausCnnBlock.forward, FixedAarryIPDnet.py:61-73.  x [nb, Cin, F, T] -> [nb, Cout, F, T // 12].
    bf16=True: weights and each conv's input rounded to bf16 (MFMA operands), fp32 accumulation / pooling / tensors."""
    wq = (lambda a: bf16_round(a)) if bf16 else (lambda a: a)   # weights AND conv inputs enter the MFMA as bf16
    out = np.maximum(conv3x3_pad12(wq(x), wq(sd[prefix + "conv1.weight"])), 0)[:, :, :, :-2]
    out = avgpool_t(out, 3)
    out = np.maximum(conv3x3_pad12(wq(out), wq(sd[prefix + "conv2.weight"])), 0)[:, :, :, :-2]
    out = avgpool_t(out, 4)
    out = conv3x3_pad12(wq(out), wq(sd[prefix + "conv3.weight"]))[:, :, :, :-2]
    return np.tanh(out, dtype=F32)


def ipdnet_forward(sd, x, is_online=True, n_seg=0, bf16=False):
    """IPDnet.forward, FixedAarryIPDnet.py:91-120.  x [nb, 2*nch, nf, nt] -> [nb, nt // 12, 2*nf, nch - 1, max_track].
    n_seg > 0 (offline networks only) = chunk-wise inference (offline_inference=True, :96-100, :114-116):
    the time axis is zero-padded to a multiple of n_seg and the segments are processed independently."""
    x = np.transpose(np.asarray(x, dtype=F32), (0, 3, 2, 1))                  # [nb, nt, nf, C]
    nb, nt, nf, _ = x.shape
    if n_seg > 0 and not is_online:
        ou_frame = nt // 12
        pad = (n_seg - nt % n_seg) % n_seg                                    # utils_.pad_segments :152-159
        xp = np.concatenate([x, np.zeros((nb, pad, nf, x.shape[3]), dtype=F32)], axis=1)
        nseg = (nt + pad) // n_seg
        xs = xp.reshape(nb * nseg, n_seg, nf, -1)                             # :98-99
        ys = ipdnet_forward(sd, np.transpose(xs, (0, 3, 2, 1)), is_online, bf16=bf16)   # [nb*nseg, nt2, 2nf, P, 2]
        nt2 = n_seg // 12
        # undo the final permute, regroup segments along time (:115), redo it
        c = np.transpose(ys, (0, 1, 4, 2, 3)).reshape(nb, nseg * nt2, 2, 2 * nf, -1)
        return np.ascontiguousarray(np.transpose(c, (0, 1, 3, 4, 2))[:, :ou_frame])
    if bf16:   # config 3: bf16 parameters (biases too, the module is .bfloat16()) and a bf16 input
        sd = {k: bf16_round(v) for k, v in sd.items()}
        x = bf16_round(x)
    y = ipdnet_block(sd, "block_1.", x, x, is_online, bf16)
    y = ipdnet_block(sd, "block_2.", y, x, is_online, bf16)
    y = np.transpose(y, (0, 3, 2, 1))                                         # [nb, C', nf, nt]
    nt2 = nt // 12
    c = caus_cnn_block(sd, "conv.", y, bf16)                                  # [nb, Cout, nf, nt2]
    c = np.transpose(c, (0, 3, 2, 1)).reshape(nb, nt2, nf, 2, -1)             # :113
    c = np.transpose(c, (0, 1, 3, 2, 4))                                      # [nb, nt2, 2, nf, K]
    return np.ascontiguousarray(np.transpose(c.reshape(nb, nt2, 2, nf * 2, -1), (0, 1, 3, 4, 2)))   # :118
