"""Torch-tensor front of the C-ABI library: PyTorch supplies device memory and
the current HIP stream, every numeric op is a HIP kernel in libfnssl_hip.so.

No function here computes on the CPU or with ATen kernels; non-ROCm tensors are
rejected (``RuntimeError``), mirroring how the reference surfaces shape errors.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np
import torch

from . import _lib
from ._lib import LstmBwdDesc, CH_MODE, LstmDesc, Net, View, check

SEG_FRAMES = 12
NBIN = 257
NF = 256


def _need_dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("fnssl: expected a ROCm device tensor (this path has no CPU implementation), got %s"
                               % (t.device if isinstance(t, torch.Tensor) else type(t)))
        if t.dtype != torch.float32:
            raise RuntimeError("fnssl: expected float32, got %s" % t.dtype)


def _first_device(objs):
    for a in objs:
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return a.device
        elif isinstance(a, (list, tuple)):
            d = _first_device(a)
            if d is not None:
                return d
    return None


def on_device(fn):
    """Run ``fn`` with the HIP current device set to the device of its first ROCm tensor argument, so that the
    kernels launch on that device and ``_stream()`` is that device's current stream (a model on ``cuda:1`` must
    not launch on device 0's stream).  Without a device tensor the call goes through unchanged (and the op's own
    ``_need_dev`` raises)."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = _first_device(args) or _first_device(kwargs.values())
        if dev is None:
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _need_dev_act(*tensors):
    """Activation tensors of the wide bf16 path: ROCm tensors of float32 or bfloat16."""
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("fnssl: expected a ROCm device tensor (this path has no CPU implementation), got %s"
                               % (t.device if isinstance(t, torch.Tensor) else type(t)))
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("fnssl: expected float32 or bfloat16, got %s" % t.dtype)


def _stream():
    """The current HIP stream of the current device (ops run under ``on_device``)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# --------------------------------------------------------------------------- #
# front end
# --------------------------------------------------------------------------- #
def num_frames(ns: int, hop: int = 256, center: bool = False) -> int:
    if hop == 256 and not center:
        return _lib.load().fnssl_num_frames(int(ns))
    return _lib.load().fnssl_num_frames_ex(int(ns), int(hop), int(bool(center)))


def num_pairs(nch: int, ch_mode: str) -> int:
    return _lib.load().fnssl_num_pairs(int(nch), CH_MODE[ch_mode])


@on_device
def stft(sig: torch.Tensor, hop: int = 256, center: bool = False):
    """sig [nb, ns, nch] -> (spec [nb, nch, nt, 257, 2], magsum [nb, nch, nt]).  Module.py:48-68; with hop 320 and
    center=True (reflect padding) it is IPDnet2's transform, IPDnet2/Module.py:47-64."""
    _need_dev(sig)
    if sig.ndim != 3:
        raise RuntimeError("fnssl.stft: expected [nb, ns, nch], got %s" % (tuple(sig.shape),))
    nb, ns, nch = sig.shape          # any strides: a permuted [nb, nch, ns] batch is read in place
    nt = num_frames(ns, hop, center)
    spec = torch.empty((nb, nch, max(nt, 0), NBIN, 2), dtype=torch.float32, device=sig.device)
    magsum = torch.empty((nb, nch, max(nt, 0)), dtype=torch.float32, device=sig.device)
    sb, sn, sc = sig.stride()
    check(_lib.load().fnssl_stft_ex(_ptr(sig), nb, ns, nch, sb, sn, sc, int(hop), int(bool(center)), _ptr(spec),
                                    _ptr(magsum), _stream()), "stft")
    return spec, magsum


_coef_cache = {}


def forgetting_coefs_host(nt: int, sample_length: int = 298):
    a = np.empty(nt, dtype=np.float32)
    b = np.empty(nt, dtype=np.float32)
    check(_lib.load().fnssl_forgetting_coefs(nt, sample_length, a.ctypes.data_as(C.c_void_p),
                                             b.ctypes.data_as(C.c_void_p)), "forgetting_coefs")
    return a, b


def forgetting_coefs(nt: int, sample_length: int, device):
    key = (nt, sample_length, str(device))
    if key not in _coef_cache:
        a, b = forgetting_coefs_host(nt, sample_length)
        _coef_cache[key] = (torch.from_numpy(a).to(device), torch.from_numpy(b).to(device))
    return _coef_cache[key]


@on_device
def pair_features(spec, magsum, ch_mode: str = "MM", eps: float = 1e-6, sample_length: int = 298,
                  layout: int = 0, normalise: bool = True):
    """main.py:207-225.  Returns (x, mu); x is [nb', nt, 256, 4] (layout 0) or [nb', 4, 256, nt] (layout 1).
    ``normalise=False`` = the reference's ``nor_flag=False`` branch (main.py:219-221: real / imag parts as they are): the same
    kernels with a zero recursion (mu = 0) and eps = 1, i.e. an exact division by one."""
    _need_dev(spec, magsum)
    nb, nch, nt = magsum.shape
    np_ = num_pairs(nch, ch_mode)
    if np_ <= 0:
        raise RuntimeError("fnssl.pair_features: need at least 2 channels, got %d" % nch)
    ca, cb = forgetting_coefs(nt, sample_length, spec.device)
    if not normalise:
        ca, cb, eps = torch.zeros_like(ca), torch.zeros_like(cb), 1.0
    mu = torch.empty((nb * np_, nt), dtype=torch.float32, device=spec.device)
    shape = (nb * np_, nt, NF, 4) if layout == 0 else (nb * np_, 4, NF, nt)
    x = torch.empty(shape, dtype=torch.float32, device=spec.device)
    check(_lib.load().fnssl_pair_features(_ptr(spec), _ptr(magsum), _ptr(ca), _ptr(cb), nb, nch, nt,
                                          CH_MODE[ch_mode], eps, _ptr(mu), _ptr(x), layout, _stream()),
          "pair_features")
    return x, mu


def preprocess(sig, ch_mode: str = "MM", eps: float = 1e-6, sample_length: int = 298, layout: int = 0, normalise: bool = True):
    """Waveforms [nb, ns, nch] -> network features (data_preprocess, main.py:200-225)."""
    spec, magsum = stft(sig)
    x, _ = pair_features(spec, magsum, ch_mode, eps, sample_length, layout, normalise)
    return x


@on_device
def array_features(spec, magsum, eps: float = 1e-6, sample_length: int = 280, layout: int = 1):
    """IPDnet's all-channel features (runIPDnetOn.py:246-254).  Returns (x, mu); x is
    [nb, 2*nch, 256, nt] (layout 1, the reference's tensor) or [nb, nt, 256, 2*nch] (layout 0)."""
    _need_dev(spec, magsum)
    nb, nch, nt = magsum.shape
    ca, cb = forgetting_coefs(nt, sample_length, spec.device)
    mu = torch.empty((nb, nt), dtype=torch.float32, device=spec.device)
    shape = (nb, nt, NF, 2 * nch) if layout == 0 else (nb, 2 * nch, NF, nt)
    x = torch.empty(shape, dtype=torch.float32, device=spec.device)
    check(_lib.load().fnssl_array_features(_ptr(spec), _ptr(magsum), _ptr(ca), _ptr(cb), nb, nch, nt, eps,
                                           _ptr(mu), _ptr(x), layout, _stream()), "array_features")
    return x, mu


def preprocess_array(sig, eps: float = 1e-6, sample_length: int = 280, layout: int = 1, hop: int = 256,
                     center: bool = False):
    """Waveforms [nb, ns, nch] -> IPDnet input features (runIPDnetOn.py:237-254); with ``hop=320, center=True,
    sample_length=249`` the features of IPDnet2 (IPDnet2/run_IPDnet2.py:277-288, see ``preprocess_ipdnet2``).

    layout 1 returns the reference's tensor [nb, 2*nch, 256, nt] as a VIEW of the frame-major storage
    [nb, nt, 256, 2*nch] the network consumes (same shape and values for every reader; ``IPDnet.forward`` then skips
    its own transposition pass, and the front end writes whole rows instead of one float per 1200-byte stride).
    ``array_features(..., layout=1)`` still produces the contiguous NCHW tensor."""
    spec, magsum = stft(sig, hop, center)
    x, _ = array_features(spec, magsum, eps, sample_length, 0)
    return x if layout == 0 else x.permute(0, 3, 2, 1)


@on_device
def array_frontend(sig, eps: float = 1e-6, sample_length: int = 280, hop: int = 256, center: bool = False):
    """Waveforms [nb, ns, nch] (any strides) -> features [nb, nt, 256, 2*nch] in ONE library call that never writes the
    spectrum (``fnssl_array_frontend``: magnitude pass, recursive mean, transform + normalise + store pass) — the
    low-memory variant of ``preprocess_array(layout=0)``: no [nb, nch, nt, 257] complex buffer (493 MB at BASELINE
    config 5), same values.  It transforms every frame twice, and the transform (VALU + LDS bound), not the traffic, is
    what the front end costs: 0.92 ms against 0.74 ms for the two-step path at config 5, so the default stays two-step."""
    _need_dev(sig)
    if sig.ndim != 3:
        raise RuntimeError("fnssl.array_frontend: expected [nb, ns, nch], got %s" % (tuple(sig.shape),))
    nb, ns, nch = sig.shape
    nt = num_frames(ns, hop, center)
    if nt <= 0:
        raise RuntimeError("fnssl.array_frontend: signal of %d samples is too short" % ns)
    ca, cb = forgetting_coefs(nt, sample_length, sig.device)
    magsum = torch.empty((nb, nch, nt), dtype=torch.float32, device=sig.device)
    mu = torch.empty((nb, nt), dtype=torch.float32, device=sig.device)
    x = torch.empty((nb, nt, NF, 2 * nch), dtype=torch.float32, device=sig.device)
    sb, sn, sc = sig.stride()
    check(_lib.load().fnssl_array_frontend(_ptr(sig), nb, ns, nch, sb, sn, sc, int(hop), int(bool(center)), _ptr(ca), _ptr(cb),
                                           eps, _ptr(magsum), _ptr(mu), _ptr(x), _stream()), "array_frontend")
    return x


def preprocess_ipdnet2(sig, eps: float = 1e-6, sample_length: int = 249, frame_major: bool = False):
    """Waveforms [nb, ns, nch] -> IPDnet2's network input [nb, 2*nch, 256, nt], nt = ns // 320 + 1
    (IPDnet2/run_IPDnet2.py:277-288: STFT nfft 512 / hop 320 / center=True (IPDnet2/Module.py:47-64), abs,
    forgetting_norm over ALL channels with sample_length 249, real / imag normalise, cat, DC drop).  The returned
    tensor has the reference's shape and values, contiguous like the reference's (frames fastest: the layout the
    encoder kernel's frame-per-lane loads coalesce on); ``frame_major=True`` returns the same tensor as a view of
    [nb, nt, 256, 2*nch] storage instead."""
    if frame_major:
        return preprocess_array(sig, eps, sample_length, 1, hop=320, center=True)
    spec, magsum = stft(sig, 320, True)
    return array_features(spec, magsum, eps, sample_length, 1)[0]   # the reference's contiguous [nb, 2 nch, 256, nt]


@on_device
def nchw_to_seq(x):
    """[n, c, nf, nt] -> [n, nt, nf, c] (Model.py:73) as a contiguous tensor."""
    _need_dev(x)
    xs = x.permute(0, 3, 2, 1)
    if xs.is_contiguous():          # already stored frame-major (e.g. what preprocess_array hands over): nothing to move
        return xs
    x = x.contiguous()
    n, c, nf, nt = x.shape
    y = torch.empty((n, nt, nf, c), dtype=torch.float32, device=x.device)
    check(_lib.load().fnssl_nchw_to_seq(_ptr(x), n, c, nf, nt, _ptr(y), _stream()), "nchw_to_seq")
    return y


# --------------------------------------------------------------------------- #
# LSTM
# --------------------------------------------------------------------------- #
def pack_lstm_host(w_ih, w_hh, b_ih, b_hh, c0: int, c2: int) -> np.ndarray:
    """Pack one direction's nn.LSTM parameters (numpy / CPU tensors) into the kernel's weight stream."""
    arrs = [np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
            for a in (w_ih, w_hh, b_ih, b_hh)]
    hidden = arrs[1].shape[1]
    if arrs[0].shape != (4 * hidden, c0 + c2) or arrs[1].shape != (4 * hidden, hidden):
        raise RuntimeError("fnssl.pack_lstm: weight shapes %s / %s do not match c0+c2=%d, H=%d"
                           % (arrs[0].shape, arrs[1].shape, c0 + c2, hidden))
    lib = _lib.load()
    n = lib.fnssl_lstm_packed_floats(c0, c2, hidden)
    if n == 0:
        raise RuntimeError("fnssl.pack_lstm: unsupported sizes c0=%d c2=%d H=%d (multiples of 4 / 16)" % (c0, c2, hidden))
    out = np.empty(n, dtype=np.float32)
    check(lib.fnssl_lstm_pack(*[a.ctypes.data_as(C.c_void_p) for a in arrs], c0, c2, hidden,
                              out.ctypes.data_as(C.c_void_p)), "lstm_pack")
    return out


def pack_lstm(w_ih, w_hh, b_ih, b_hh, c0: int, c2: int, device) -> torch.Tensor:
    return torch.from_numpy(pack_lstm_host(w_ih, w_hh, b_ih, b_hh, c0, c2)).to(device)


def pack_lstm_bf16(w_ih, w_hh, b_ih, b_hh, c0: int, c2: int, device) -> torch.Tensor:
    """Weight stream of the bf16-MFMA kernels: weights rounded to bf16 (nearest even), bias fp32.
    c0, c2 multiples of 16, hidden a multiple of 32."""
    arrs = [np.ascontiguousarray(a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
            for a in (w_ih, w_hh, b_ih, b_hh)]
    hidden = arrs[1].shape[1]
    if arrs[0].shape != (4 * hidden, c0 + c2) or arrs[1].shape != (4 * hidden, hidden):
        raise RuntimeError("fnssl.pack_lstm_bf16: weight shapes %s / %s do not match c0+c2=%d, H=%d"
                           % (arrs[0].shape, arrs[1].shape, c0 + c2, hidden))
    lib = _lib.load()
    n = lib.fnssl_lstm_packed_floats_bf16(c0, c2, hidden)
    if n == 0:
        raise RuntimeError("fnssl.pack_lstm_bf16: unsupported sizes c0=%d c2=%d H=%d (16 / 16 / 32)" % (c0, c2, hidden))
    out = np.empty(n, dtype=np.float32)
    check(lib.fnssl_lstm_pack_bf16(*[a.ctypes.data_as(C.c_void_p) for a in arrs], c0, c2, hidden,
                                   out.ctypes.data_as(C.c_void_p)), "lstm_pack_bf16")
    return torch.from_numpy(out).to(device)


def pack_lstm_bf16w(w_ih, w_hh, b_ih, b_hh, c0: int, c2: int, device) -> torch.Tensor:
    """Weight stream of the WIDE bf16 kernels (32 sequences per wave, lstm_bf16w.h): gate-row tiles of bf16
    A operands, the bias as three bf16 terms of a constant-one block.  c0, c2, hidden multiples of 16."""
    arrs = [np.ascontiguousarray(a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
            for a in (w_ih, w_hh, b_ih, b_hh)]
    hidden = arrs[1].shape[1]
    if arrs[0].shape != (4 * hidden, c0 + c2) or arrs[1].shape != (4 * hidden, hidden):
        raise RuntimeError("fnssl.pack_lstm_bf16w: weight shapes %s / %s do not match c0+c2=%d, H=%d"
                           % (arrs[0].shape, arrs[1].shape, c0 + c2, hidden))
    lib = _lib.load()
    n = lib.fnssl_lstm_packed_floats_bf16w(c0, c2, hidden)
    if n == 0:
        raise RuntimeError("fnssl.pack_lstm_bf16w: unsupported sizes c0=%d c2=%d H=%d (multiples of 16)" % (c0, c2, hidden))
    out = np.empty(n, dtype=np.float32)
    check(lib.fnssl_lstm_pack_bf16w(*[a.ctypes.data_as(C.c_void_p) for a in arrs], c0, c2, hidden,
                                    out.ctypes.data_as(C.c_void_p)), "lstm_pack_bf16w")
    return torch.from_numpy(out).to(device)


def _view(t, mode):
    """4-D logical [nb, nt, nf, C] tensor -> (View, q_inner) for 'full' (seq=(b,t), step=f) or
    'narrow' (seq=(b,f), step=t)."""
    sb, st, sf, sc = t.stride()
    if sc != 1:
        raise RuntimeError("fnssl.lstm: channel dimension must be contiguous")
    if mode == "full":
        return View(t.data_ptr(), sb, st, sf)
    return View(t.data_ptr(), sb, sf, st)


def _conform(t):
    """Channel-contiguous, 16-byte aligned, strides multiples of 4 floats (8 bf16) — else copy (plumbing only)."""
    if t is None:
        return None
    m = 8 if t.dtype == torch.bfloat16 else 4
    ok = t.stride(-1) == 1 and t.data_ptr() % 16 == 0 and all(s % m == 0 for s in t.stride()[:-1])
    return t if ok else t.contiguous()


_ws_cache = {}
_fb_counters = {}
_fb_nets = None      # weak set of DeviceNet objects (each owns the counter its fnssl_net hands to the library)


def fallback_counter(device):
    """The per-device int32 counter every LSTM call of this binding hands to the library unless the caller passes its own
    (fnssl_lstm_desc.fallback_count / fnssl_lstm_bwd_desc.fallback_count): launches whose cluster-resident kernel gave up on
    a hand-off and were recomputed by the guarded fallback kernels of the same call.  Results are correct either way; a
    non-zero count means the device could not keep a cluster's member workgroups resident (shared, partitioned or busy
    device) and throughput numbers taken meanwhile are not the kernels'."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    c = _fb_counters.get(device.index)
    if c is None:
        c = torch.zeros(1, dtype=torch.int32, device=device)
        _fb_counters[device.index] = c
    return c


def cluster_fallbacks(device=None, reset: bool = False) -> int:
    """Fallbacks counted on ``device`` so far: ``fallback_counter(device)`` plus the counters of the live ``DeviceNet``s
    (SYNCHRONISES the device: measurement code calls it outside timed regions).  ``reset``: zero them afterwards."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    cs = [fallback_counter(device)] + [n.fallbacks for n in (_fb_nets or ()) if n.device == device or
                                       (n.device.index is None and device.index == torch.cuda.current_device())]
    total = sum(int(c.item()) for c in cs)
    if reset:
        for c in cs:
            c.zero_()
        for n in (_fb_nets or ()):
            n._fb_seen = 0
    return total



def _workspace(nbytes: int, device, tag: str):
    """Scratch buffer for one op, cached per (device, stream, tag): two forwards issued on different HIP streams
    never share scratch, and on one stream the ops that use a tag are ordered anyway.  A buffer that has to
    grow is replaced; the old one goes back to torch's caching allocator, which keeps it reserved for this
    stream's pending work."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def release_workspaces():
    _ws_cache.clear()


@on_device
def lstm_layer(mode: str, x0, x1, x2, packed, hidden: int, out, variant: int = 0, skip=None, out_sum=None,
               reserve=None, carry_workspace=None, carry=False, bf16=False, wide=False, fallback_count=None, plan_only=False,
               tuning=None):
    """One (bi)LSTM layer over strided views.

    ``fallback_count`` (optional 1-element int32 device tensor, caller-zeroed): counts the layers whose cluster-resident
    kernel gave up on a hand-off and were recomputed by the guarded fallback kernels of the same call (results are
    correct either way).  ``tuning`` (a ``_lib.Tuning``, e.g. ``_lib.make_tuning(reserved_cus=16)``): this call's knobs
    instead of the process default (fnssl_lstm_desc.tuning).  ``plan_only=True`` launches nothing and returns ``(family, rounds)`` — the kernel family
    ``fnssl_lstm_forward`` takes for exactly this call (``_lib.LSTM_FAMILY`` names), see ``lstm_plan``.

    mode 'full': sequences are (b, t) rows, steps run over f; 'narrow': sequences (b, f), steps over t.
    x0 (+ x1) is the summed input, x2 the concatenated one (either may be None); all are logical
    [nb, nt, nf, C] tensors with arbitrary strides.  ``packed`` is a list of 1 or 2 device weight
    streams; ``out`` a logical [nb, nt, nf, ndir*hidden] tensor (any strides) that is written in place.
    With ``skip``/``out_sum`` (both logical [nb, nt, nf, ndir*hidden]; out_sum with out's strides) the
    kernel also stores out_sum = h + skip, i.e. the next layer's residual input.
    ``reserve`` (a float32 device buffer of ``lstm_reserve_floats`` elements) switches to the training
    forward, which also saves the gate activations and cell states for ``lstm_backward``.
    Streaming (uni-directional layers): pass a persistent ``carry_workspace`` (``lstm_state_workspace``); with
    ``carry=True`` the recurrence continues from the cell state in it and from the h row one step before
    ``out`` in memory (the caller's buffer holds the previous call's last h there).
    ``bf16=True``: ``packed`` comes from ``pack_lstm_bf16`` and the matrix product runs on bf16 MFMAs (operands
    rounded to bf16, fp32 accumulate; all tensors stay fp32).
    ``wide=True`` (with bf16): the 32-sequences-per-wave kernels — ``packed`` from ``pack_lstm_bf16w``; x0 / x2 /
    out may each be float32 or bfloat16 tensors (bf16 between layers: the values are rounded to bf16 on entry to
    the next MFMA anyway), no x1 / skip / reserve / carry.
    """
    if wide:
        _need_dev_act(x0, x2, out)
        _need_dev(x1, skip, out_sum, *packed)
    else:
        _need_dev(x0, x1, x2, out, skip, out_sum, *packed)
    if mode not in ("full", "narrow"):
        raise RuntimeError("fnssl.lstm_layer: mode must be 'full' or 'narrow'")
    ref = x0 if x0 is not None else x2
    if ref is None:
        raise RuntimeError("fnssl.lstm_layer: no input")
    x0, x1, x2 = _conform(x0), _conform(x1), _conform(x2)
    nb, nt, nf = ref.shape[:3]
    ndir = len(packed)
    if tuple(out.shape) != (nb, nt, nf, ndir * hidden):
        raise RuntimeError("fnssl.lstm_layer: out shape %s != %s" % (tuple(out.shape), (nb, nt, nf, ndir * hidden)))
    d = LstmDesc()
    if x0 is not None:
        d.src0 = _view(x0, mode)
        d.c0 = x0.shape[3]
    if x1 is not None:
        if x0 is None or x1.shape != x0.shape:
            raise RuntimeError("fnssl.lstm_layer: x1 must match x0")
        d.src1 = _view(x1, mode)
    if x2 is not None:
        d.src2 = _view(x2, mode)
        d.c2 = x2.shape[3]
    ov = _view(out, mode)
    d.out, d.out_so, d.out_si, d.out_st = ov.p, ov.so, ov.si, ov.st
    if (skip is None) != (out_sum is None):
        raise RuntimeError("fnssl.lstm_layer: skip and out_sum go together")
    if out_sum is not None:
        if out_sum.shape != out.shape or out_sum.stride() != out.stride() or skip.shape != out.shape:
            raise RuntimeError("fnssl.lstm_layer: out_sum must have out's shape and strides, skip out's shape")
        skip = _conform(skip)
        d.skip = _view(skip, mode)
        d.out_sum = out_sum.data_ptr()
    d.hidden, d.ndir = hidden, ndir
    d.nseq = nb * (nt if mode == "full" else nf)
    d.q_inner = nt if mode == "full" else nf
    d.nsteps = nf if mode == "full" else nt
    d.wpack[0] = packed[0].data_ptr()
    d.wpack[1] = packed[1].data_ptr() if ndir == 2 else 0
    lib = _lib.load()
    precision = (2 if wide else 1) if bf16 else 0
    wsb = lib.fnssl_lstm_workspace_bytes_ex(d.nseq, hidden, ndir, precision)
    if carry_workspace is not None:
        _need_dev(carry_workspace)
        if carry_workspace.numel() * 4 < wsb:
            raise RuntimeError("fnssl.lstm_layer: carry workspace too small")
        d.workspace, d.workspace_bytes = carry_workspace.data_ptr(), carry_workspace.numel() * 4
        d.carry_state = 1 if carry else 0
    else:
        if carry:
            raise RuntimeError("fnssl.lstm_layer: carry=True needs the persistent carry_workspace")
        ws = _workspace(wsb, out.device, "lstm")
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    d.variant = variant
    d.precision = precision
    if wide:
        if not bf16:
            raise RuntimeError("fnssl.lstm_layer: wide=True is a bf16 mode")
        d.f32_mask = ((1 if (x0 is not None and x0.dtype == torch.float32) else 0) |
                      (2 if (x2 is not None and x2.dtype == torch.float32) else 0) |
                      (4 if out.dtype == torch.float32 else 0))
    if reserve is not None:
        _need_dev(reserve)
        d.reserve, d.reserve_bytes = reserve.data_ptr(), reserve.numel() * 4
    if fallback_count is None:
        fallback_count = fallback_counter(out.device)
    if not isinstance(fallback_count, torch.Tensor) or not fallback_count.is_cuda or fallback_count.dtype != torch.int32 \
            or fallback_count.numel() < 1:
        raise RuntimeError("fnssl.lstm_layer: fallback_count must be an int32 device tensor")
    d.fallback_count = fallback_count.data_ptr()
    if tuning is not None:
        d.tuning = C.pointer(tuning)
    if plan_only:
        fam, rounds = C.c_int(0), C.c_int(0)
        check(lib.fnssl_lstm_plan(C.byref(d), C.byref(fam), C.byref(rounds)), "lstm_plan")
        return _lib.LSTM_FAMILY.get(int(fam.value), "unknown(%d)" % fam.value), int(rounds.value)
    check(lib.fnssl_lstm_forward(C.byref(d), _stream()), "lstm_forward")
    return out


def lstm_plan(*args, **kwargs):
    """``(family, rounds)`` of the ``lstm_layer`` call with the same arguments: which kernel family the library would
    launch on this device with the current environment, without launching anything (fnssl_lstm_plan)."""
    kwargs["plan_only"] = True
    return lstm_layer(*args, **kwargs)


def mfma_f32_peak(iters: int = 20000, waves_per_simd: int = 2, reps: int = 3, device=None) -> float:
    """The device's own fp32-MFMA ceiling in TFLOP/s (fnssl_mfma_f32_peak timed with HIP events on the current stream;
    best of ``reps``): the calibration bench.py reports as roofline.peak_measured."""
    lib = _lib.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    out = torch.empty(ncu * waves_per_simd * 256, dtype=torch.float32, device=device)
    best = 0.0
    for r in range(reps + 1):
        flop = C.c_double(0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.fnssl_mfma_f32_peak(out.data_ptr(), out.numel(), iters, waves_per_simd, C.byref(flop), _stream()), "mfma_f32_peak")
        e1.record()
        e1.synchronize()
        if r > 0:   # (the first launch carries the code-object load)
            best = max(best, flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def mfma_f32_peak_sustained(seconds: float = 2.0, iters: int = 20000, waves_per_simd: int = 2, device=None) -> dict:
    """The same ceiling held for ``seconds`` of back-to-back launches (fnssl_mfma_f32_peak_clocks): what the matrix pipe
    delivers once the power / thermal management has settled, which is the regime the 85 – 115 ms LSTM kernels of a 10-s
    timed region run in — a 35-ms burst does not show it.  Returns TFLOP/s of the whole window and of its last quarter,
    and the clock (MHz, shader cycles per 100 MHz tick of wave 0 of every workgroup) per XCD during the LAST launch."""
    lib = _lib.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    nblk = ncu * waves_per_simd
    out = torch.empty(nblk * 256, dtype=torch.float32, device=device)
    clk = torch.zeros(nblk * 3, dtype=torch.int64, device=device)
    flop = C.c_double(0.0)

    def launch(with_clocks):
        check(lib.fnssl_mfma_f32_peak_clocks(out.data_ptr(), out.numel(), iters, waves_per_simd, C.byref(flop),
                                             clk.data_ptr() if with_clocks else None, clk.numel() if with_clocks else 0,
                                             _stream()), "mfma_f32_peak_clocks")

    launch(False)                                       # code-object load + one-launch duration
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(False)
    e1.record()
    e1.synchronize()
    one_ms = max(e0.elapsed_time(e1), 1e-3)
    n = max(4, int(seconds * 1e3 / one_ms + 0.999))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        launch(k == n - 1)
        ev[k + 1].record()
    ev[n].synchronize()
    total_ms = ev[0].elapsed_time(ev[n])
    q = max(1, n // 4)
    last_ms = ev[n - q].elapsed_time(ev[n])
    c = clk.cpu().numpy().reshape(nblk, 3)
    mhz = c[:, 0] / np.maximum(c[:, 1], 1) * 100.0
    per_xcd = {}
    for x in sorted(set(int(v) for v in c[:, 2])):
        m = mhz[c[:, 2] == x]
        per_xcd[x] = round(float(m.mean()), 1)
    return {"tflops": flop.value * n / (total_ms * 1e-3) / 1e12, "tflops_last_quarter": flop.value * q / (last_ms * 1e-3) / 1e12,
            "seconds": total_ms * 1e-3, "launches": n, "xcd_mhz": per_xcd,
            "slowest_xcd_mhz": min(per_xcd.values()), "fastest_xcd_mhz": max(per_xcd.values())}


def lstm_cluster_status(nseq: int, hidden: int, ndir: int, device=None) -> int:
    """Status word the cluster-resident LSTM kernels left in the current stream's LSTM workspace (0 = every hand-off
    arrived; otherwise the layer was recomputed by the guarded fallback kernels of the same call — see
    fnssl_lstm_cluster_status).  `nseq, hidden, ndir` as in the lstm_layer call that used the workspace."""
    lib = _lib.load()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ws = _workspace(lib.fnssl_lstm_workspace_bytes_ex(nseq, hidden, ndir, 0), device, "lstm")   # (the cached buffer: never shrinks)
    out = C.c_uint(0xffffffff)
    check(lib.fnssl_lstm_cluster_status(ws.data_ptr(), ws.numel(), nseq, hidden, ndir, _stream(), C.byref(out)),
          "lstm_cluster_status")
    return int(out.value)


def lstm_state_workspace(nseq: int, hidden: int, device):
    """Persistent per-layer cell-state buffer for streaming (uni-directional) LSTM calls."""
    n = _lib.load().fnssl_lstm_workspace_bytes_ex(nseq, hidden, 1, 0)
    return torch.zeros((n + 3) // 4, dtype=torch.float32, device=device)


def lstm_reserve_floats(nseq: int, hidden: int, ndir: int, nsteps: int) -> int:
    return _lib.load().fnssl_lstm_reserve_bytes(nseq, hidden, ndir, nsteps) // 4


def pack_lstm_bwd_host(w_ih, w_hh, c0g: int) -> np.ndarray:
    """Pack one direction's [W_ih[:, :c0g] | W_hh]^T into the backward kernel's weight stream."""
    arrs = [np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
            for a in (w_ih, w_hh)]
    hidden = arrs[1].shape[1]
    c_in = arrs[0].shape[1]
    lib = _lib.load()
    n = lib.fnssl_lstm_bwd_packed_floats(c0g, hidden)
    if n == 0 or arrs[0].shape[0] != 4 * hidden or arrs[1].shape[0] != 4 * hidden:
        raise RuntimeError("fnssl.pack_lstm_bwd: unsupported sizes c0g=%d H=%d" % (c0g, hidden))
    out = np.empty(n, dtype=np.float32)
    check(lib.fnssl_lstm_pack_bwd(arrs[0].ctypes.data_as(C.c_void_p), arrs[1].ctypes.data_as(C.c_void_p), c_in, c0g,
                                  hidden, out.ctypes.data_as(C.c_void_p)), "lstm_pack_bwd")
    return out


@on_device
def lstm_backward(mode: str, reserve, dh, da, dx, packed_bwd, hidden: int, c0g: int, plan_only: bool = False,
                  status: bool = False, fallback_count=None, tuning=None):
    """Back-propagation through time of one (bi)LSTM layer.

    dh: upstream gradient, logical [nb, nt, nf, ndir*hidden]; da (written): logical [nb, nt, nf, ndir*4*hidden]
    pre-activation gate gradients; dx (written, or None when c0g == 0): logical [nb, nt, nf, ndir*c0g], one slab
    per direction.  ``mode`` as in lstm_layer.  ``plan_only``: launch nothing, return the kernel family name the call
    would take (fnssl_lstm_backward_plan); ``status``: also return the cluster kernel's status word (synchronises).
    ``fallback_count`` (int32 device tensor, caller-zeroed): incremented when the cluster-resident BPTT kernel gave up on a
    hand-off and the split kernels of the same call recomputed the layer; ``tuning``: this call's knobs (a ``_lib.Tuning``)."""
    _need_dev(reserve, dh, da, dx, *packed_bwd)
    ndir = len(packed_bwd)
    nb, nt, nf = dh.shape[:3]
    if tuple(dh.shape) != (nb, nt, nf, ndir * hidden) or tuple(da.shape) != (nb, nt, nf, ndir * 4 * hidden):
        raise RuntimeError("fnssl.lstm_backward: dh / da shapes %s / %s" % (tuple(dh.shape), tuple(da.shape)))
    if (c0g > 0) != (dx is not None) or (dx is not None and tuple(dx.shape) != (nb, nt, nf, ndir * c0g)):
        raise RuntimeError("fnssl.lstm_backward: dx must be [nb, nt, nf, ndir*c0g] (None when c0g == 0)")
    dh = _conform(dh)
    d = LstmBwdDesc()
    d.reserve = reserve.data_ptr()
    d.dh = _view(dh, mode)
    v = _view(da, mode)
    d.da, d.da_so, d.da_si, d.da_st = v.p, v.so, v.si, v.st
    if dx is not None:
        v = _view(dx, mode)
        d.dx, d.dx_so, d.dx_si, d.dx_st = v.p, v.so, v.si, v.st
    d.c0g, d.hidden, d.ndir = c0g, hidden, ndir
    d.nseq = nb * (nt if mode == "full" else nf)
    d.q_inner = nt if mode == "full" else nf
    d.nsteps = nf if mode == "full" else nt
    if reserve.numel() < lstm_reserve_floats(d.nseq, hidden, ndir, d.nsteps):
        raise RuntimeError("fnssl.lstm_backward: reserve buffer too small")
    d.wpack_bwd[0] = packed_bwd[0].data_ptr()
    d.wpack_bwd[1] = packed_bwd[1].data_ptr() if ndir == 2 else 0
    lib = _lib.load()
    ws = _workspace(lib.fnssl_lstm_bwd_workspace_bytes(d.nseq, hidden, ndir), dh.device, "lstm_bwd")
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if fallback_count is None:
        fallback_count = fallback_counter(dh.device)
    if not isinstance(fallback_count, torch.Tensor) or not fallback_count.is_cuda or fallback_count.dtype != torch.int32:
        raise RuntimeError("fnssl.lstm_backward: fallback_count must be an int32 device tensor")
    d.fallback_count = fallback_count.data_ptr()
    if tuning is not None:
        d.tuning = C.pointer(tuning)
    if plan_only:
        fam = C.c_int(0)
        check(lib.fnssl_lstm_backward_plan(C.byref(d), C.byref(fam)), "lstm_backward_plan")
        return _lib.LSTM_FAMILY[fam.value]
    check(lib.fnssl_lstm_backward(C.byref(d), _stream()), "lstm_backward")
    if status:
        word = C.c_uint(0)
        check(lib.fnssl_lstm_backward_status(ws.data_ptr(), ws.numel(), d.nseq, hidden, ndir, _stream(), C.byref(word)),
              "lstm_backward_status")
        return da, dx, word.value
    return da, dx


# --------------------------------------------------------------------------- #
# causal 3x3 conv + time pooling (IPDnet head)
# --------------------------------------------------------------------------- #
def pack_conv3x3(weight, ca: int, cb: int, device, bf16: bool = False) -> torch.Tensor:
    """Pack a Conv2d weight [cout, ca + cb, 3, 3] into the conv kernel's weight stream (``bf16``: the bf16-MFMA
    stream; ca, cb multiples of 16)."""
    w = np.ascontiguousarray(weight.detach().float().cpu().numpy() if isinstance(weight, torch.Tensor) else weight,
                             dtype=np.float32)
    cout = w.shape[0]
    if w.shape != (cout, ca + cb, 3, 3):
        raise RuntimeError("fnssl.pack_conv3x3: weight shape %s does not match [cout, %d + %d, 3, 3]"
                           % (w.shape, ca, cb))
    lib = _lib.load()
    n = (lib.fnssl_conv3x3_packed_floats_bf16 if bf16 else lib.fnssl_conv3x3_packed_floats)(cout, ca, cb)
    if n == 0:
        raise RuntimeError("fnssl.pack_conv3x3: unsupported sizes cout=%d ca=%d cb=%d" % (cout, ca, cb))
    out = np.empty(n, dtype=np.float32)
    check((lib.fnssl_conv3x3_pack_bf16 if bf16 else lib.fnssl_conv3x3_pack)(
        w.ctypes.data_as(C.c_void_p), cout, ca, cb, out.ctypes.data_as(C.c_void_p)), "conv3x3_pack")
    return torch.from_numpy(out).to(device)


@on_device
def conv3x3_causal(xa, xb, packed, cout: int, act: str = "none", bf16: bool = False):
    """Causal 3x3 conv over (bin, time) of the channel concatenation [xa | xb].

    xa / xb: logical [nb, nf, nt, C] tensors with arbitrary batch / bin / time strides and a
    contiguous channel dimension (xb may be None).  Returns [nb, nf, nt, ceil4(cout)] channels-last
    (padding channels are 0)."""
    _need_dev_act(xa)
    _need_dev(xb, packed)
    a_bf = xa.dtype == torch.bfloat16
    if a_bf and not bf16:
        raise RuntimeError("fnssl.conv3x3_causal: a bfloat16 input needs the bf16 kernel (bf16=True)")
    xa, xb = _conform(xa), _conform(xb)
    nb, nf, nt, ca = xa.shape
    cb = 0 if xb is None else xb.shape[3]
    if xb is not None and tuple(xb.shape[:3]) != (nb, nf, nt):
        raise RuntimeError("fnssl.conv3x3_causal: xb must match xa's [nb, nf, nt]")
    if xa.stride(3) != 1 or (xb is not None and xb.stride(3) != 1):
        raise RuntimeError("fnssl.conv3x3_causal: channel dimension must be contiguous")
    cs = (cout + 3) // 4 * 4
    out = torch.empty((nb, nf, nt, cs), dtype=torch.float32, device=xa.device)   # pad channels come out 0
    sa = xa.stride()
    sb_ = xb.stride() if xb is not None else (0, 0, 0, 1)
    code = {"none": 0, "relu": 1, "tanh": 2}[act]
    lib = _lib.load()
    fn = (lib.fnssl_conv3x3_causal_bf16a if a_bf else lib.fnssl_conv3x3_causal_bf16) if bf16 else lib.fnssl_conv3x3_causal
    check(fn(_ptr(xa), sa[0], sa[1], sa[2], ca,
             _ptr(xb) if xb is not None else None, sb_[0], sb_[1], sb_[2], cb,
             _ptr(packed), cout, nb, nf, nt, code, _ptr(out), cs, _stream()),
          "conv3x3_causal")
    return out


def conv3x3_bf16x_supported(cout: int, ca: int, cb: int) -> bool:
    """Whether the LDS-staged bf16 conv kernel (conv_bf16x.hip) takes these channel counts."""
    return _lib.load().fnssl_conv3x3_packed_bytes_bf16x(cout, ca, cb) > 0


def pack_conv3x3_bf16x(weight, ca: int, cb: int, device) -> torch.Tensor:
    """Pack a Conv2d weight [cout, ca + cb, 3, 3] into the weight stream of ``conv3x3_causal_bf16x``."""
    w = np.ascontiguousarray(weight.detach().float().cpu().numpy() if isinstance(weight, torch.Tensor) else weight,
                             dtype=np.float32)
    cout = w.shape[0]
    if w.shape != (cout, ca + cb, 3, 3):
        raise RuntimeError("fnssl.pack_conv3x3_bf16x: weight shape %s does not match [cout, %d + %d, 3, 3]"
                           % (w.shape, ca, cb))
    lib = _lib.load()
    n = lib.fnssl_conv3x3_packed_bytes_bf16x(cout, ca, cb)
    if n == 0:
        raise RuntimeError("fnssl.pack_conv3x3_bf16x: unsupported sizes cout=%d ca=%d cb=%d" % (cout, ca, cb))
    out = np.empty(n // 4, dtype=np.float32)       # raw stream bytes, carried as a float32 tensor like the others
    check(lib.fnssl_conv3x3_pack_bf16x(w.ctypes.data_as(C.c_void_p), cout, ca, cb, out.ctypes.data_as(C.c_void_p)),
          "conv3x3_pack_bf16x")
    return torch.from_numpy(out).to(device)


@on_device
def conv3x3_causal_bf16x(xa, xb, packed, cout: int, act: str = "none", pool: int = 1, bf16_out: bool = False):
    """``conv3x3_causal`` for a bfloat16 ``xa`` (and an optional fp32 ``xb``) through the LDS-staged kernel, with the
    time pooling that follows it in CausCnnBlock fused into the epilogue (``pool`` 1 / 3 / 4).
    Returns [nb, nf, nt // pool, cout], float32 or (``bf16_out``) bfloat16."""
    _need_dev_act(xa)
    _need_dev(xb, packed)
    if xa.dtype != torch.bfloat16 or (xb is not None and xb.dtype != torch.float32):
        raise RuntimeError("fnssl.conv3x3_causal_bf16x: xa must be bfloat16 and xb float32")
    nb, nf, nt, ca = xa.shape
    cb = 0 if xb is None else xb.shape[3]
    if xb is not None and tuple(xb.shape[:3]) != (nb, nf, nt):
        raise RuntimeError("fnssl.conv3x3_causal_bf16x: xb must match xa's [nb, nf, nt]")
    if xa.stride(3) != 1 or (xb is not None and xb.stride(3) != 1):
        raise RuntimeError("fnssl.conv3x3_causal_bf16x: channel dimension must be contiguous")
    out = torch.empty((nb, nf, nt // pool, cout), dtype=torch.bfloat16 if bf16_out else torch.float32, device=xa.device)
    if out.numel() == 0:
        return out
    sa = xa.stride()
    sb_ = xb.stride() if xb is not None else (0, 0, 0, 1)
    code = {"none": 0, "relu": 1, "tanh": 2}[act]
    check(_lib.load().fnssl_conv3x3_causal_bf16x(
        _ptr(xa), sa[0], sa[1], sa[2], ca, _ptr(xb) if xb is not None else None, sb_[0], sb_[1], sb_[2], cb,
        _ptr(packed), cout, nb, nf, nt, code, pool, int(bf16_out), _ptr(out), cout, _stream()), "conv3x3_causal_bf16x")
    return out


@on_device
def avgpool_time(x, k: int, bf16_out: bool = False):
    """[nb, nf, nt, C] channels-last -> [nb, nf, nt // k, C] (AvgPool2d((1, k)) of the NCHW view); ``bf16_out``
    rounds the result to bfloat16 (the operand rounding of the bf16 conv that follows, done once here)."""
    _need_dev(x)
    x = x.contiguous()
    nb, nf, nt, c = x.shape
    y = torch.empty((nb, nf, nt // k, c), dtype=torch.bfloat16 if bf16_out else torch.float32, device=x.device)
    if y.numel():
        fn = _lib.load().fnssl_avgpool_time_bf16 if bf16_out else _lib.load().fnssl_avgpool_time
        check(fn(_ptr(x), nb * nf, nt, c, k, _ptr(y), _stream()), "avgpool_time")
    return y


@on_device
def lstm_weight_grads(da, x0, x2, h, hidden: int, ndir: int, nsteps: int, g_wih, g_whh, g_bih, g_bhh):
    """Accumulate one LSTM layer's weight gradients (``fnssl_lstm_weight_grads``, csrc/wgrad.hip): g_wih[d] [4H, c0+c2]
    += dA_d^T [x0 | x2], g_whh[d] [4H, H] += dA_d^T h_prev_d, g_bih[d] = g_bhh[d] [4H] += sum_r dA_d[r].
    da [rows, ndir*4H], x0 [rows, c0] / x2 [rows, c2] (either may be None), h [rows, ndir*H]: row-major matrices with
    rows = sequences x steps in the layer's natural layout (any row stride).  g_*: lists of ndir contiguous tensors."""
    from ._lib import WgradDesc
    _need_dev(da, x0, x2, h, *g_wih, *g_whh, *g_bih, *g_bhh)
    rows = da.shape[0]
    d = WgradDesc()
    d.da, d.lda = _ptr(da), da.stride(0)
    d.x0, d.ldx0, d.c0 = (_ptr(x0), x0.stride(0), x0.shape[1]) if x0 is not None else (None, 0, 0)
    d.x2, d.ldx2, d.c2 = (_ptr(x2), x2.stride(0), x2.shape[1]) if x2 is not None else (None, 0, 0)
    d.h, d.ldh = _ptr(h), h.stride(0)
    for t in (da, x0, x2, h):
        if t is not None and (t.stride(1) != 1 or t.dtype != torch.float32):
            raise RuntimeError("fnssl.lstm_weight_grads: operands must be fp32 row-major matrices")
    if rows % nsteps or h.shape[0] != rows or da.shape[1] != ndir * 4 * hidden or h.shape[1] != ndir * hidden:
        raise RuntimeError("fnssl.lstm_weight_grads: shapes da %s / h %s do not match hidden %d, ndir %d, nsteps %d"
                           % (tuple(da.shape), tuple(h.shape), hidden, ndir, nsteps))
    d.nseq, d.nsteps, d.hidden, d.ndir = rows // nsteps, nsteps, hidden, ndir
    for k in range(ndir):
        for name, lst, shape in (("g_wih", g_wih, (4 * hidden, d.c0 + d.c2)), ("g_whh", g_whh, (4 * hidden, hidden)),
                                 ("g_bih", g_bih, (4 * hidden,)), ("g_bhh", g_bhh, (4 * hidden,))):
            t = lst[k]
            if tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32:
                raise RuntimeError("fnssl.lstm_weight_grads: %s[%d] must be a contiguous fp32 %s" % (name, k, shape))
            getattr(d, name)[k] = t.data_ptr()
    lib = _lib.load()
    ws = _workspace(lib.fnssl_lstm_weight_grads_workspace_bytes(rows, hidden, ndir, d.c0, d.c2), da.device, "wgrad")
    d.workspace, d.workspace_bytes = _ptr(ws), ws.numel()
    check(lib.fnssl_lstm_weight_grads(C.byref(d), _stream()), "lstm_weight_grads")


# --------------------------------------------------------------------------- #
# head / whole network
# --------------------------------------------------------------------------- #
@on_device
def head(x, w, b):
    """x [nb, nf, nt, 256] -> [nb, nt//12, 2*nf]  (Model.py:79-87)."""
    _need_dev(x, w, b)
    x = x.contiguous()
    nb, nf, nt, c = x.shape
    if c != 256:
        raise RuntimeError("fnssl.head: emb2ipd is Linear(256, 2); got %d channels" % c)
    out = torch.empty((nb, nt // SEG_FRAMES, 2 * nf), dtype=torch.float32, device=x.device)
    check(_lib.load().fnssl_head(_ptr(x), nb, nf, nt, _ptr(w.contiguous()), _ptr(b.contiguous()), _ptr(out),
                                 _stream()), "head")
    return out


@on_device
def linear(x, wt, b):
    """y = x @ wt + b with wt = weight^T [k, n_out]."""
    _need_dev(x, wt, b)
    x = x.contiguous()
    k = x.shape[-1]
    m = x.numel() // k
    n_out = wt.shape[1]
    y = torch.empty(tuple(x.shape[:-1]) + (n_out,), dtype=torch.float32, device=x.device)
    check(_lib.load().fnssl_linear(_ptr(x), m, k, _ptr(wt.contiguous()), _ptr(b), n_out, _ptr(y), _stream()),
          "linear")
    return y


class DeviceNet:
    """Device-resident packed parameters of an FN_SSL network (the ``fnssl_net`` struct)."""

    def __init__(self, state: dict, device, is_online: bool = True, is_doa: bool = False, input_size: int = 4,
                 prefix: str = ""):
        self.device = torch.device(device)
        self.is_online, self.is_doa, self.input_size = bool(is_online), bool(is_doa), int(input_size)
        self._keep = []
        net = Net()
        g = lambda k: state[prefix + k]  # noqa: E731
        for blk in range(3):
            bp = "block_%d." % (blk + 1)
            fin = input_size if blk == 0 else 256
            specs = [("fullLstm.", fin, 0, True),
                     ("narrLstm.", 256, input_size if blk == 0 else 0, not is_online)]
            for li, (lp, c0, c2, bidir) in enumerate(specs):
                for di, sfx in enumerate(["", "_reverse"] if bidir else [""]):
                    t = pack_lstm(g(bp + lp + "weight_ih_l0" + sfx), g(bp + lp + "weight_hh_l0" + sfx),
                                  g(bp + lp + "bias_ih_l0" + sfx), g(bp + lp + "bias_hh_l0" + sfx), c0, c2,
                                  self.device)
                    self._keep.append(t)
                    net.wpack[blk][li][di] = t.data_ptr()
        self.emb_w = self._dev(g("emb2ipd.weight"))
        self.emb_b = self._dev(g("emb2ipd.bias"))
        net.emb_w, net.emb_b = self.emb_w.data_ptr(), self.emb_b.data_ptr()
        if is_doa:
            self.doa_wt = self._dev(g("ipd2doa.weight")).t().contiguous()
            self.doa_b = self._dev(g("ipd2doa.bias"))
            net.doa_wt, net.doa_b = self.doa_wt.data_ptr(), self.doa_b.data_ptr()
        net.input_size, net.is_online = self.input_size, int(self.is_online)
        # layers whose cluster-resident kernel gave up on a hand-off and were recomputed by the guarded fallback kernels of
        # the same call (fnssl_lstm_forward): counted on the device, read back asynchronously (4 bytes per forward, no
        # synchronisation) and reported one forward later as a RuntimeWarning — results are correct either way
        self.fallbacks = torch.zeros(1, dtype=torch.int32, device=self.device)
        global _fb_nets
        if _fb_nets is None:
            import weakref
            _fb_nets = weakref.WeakSet()
        _fb_nets.add(self)
        self._fb_host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.device.type == "cuda" else None
        self._fb_event, self._fb_seen = None, 0
        net.fallback_count = self.fallbacks.data_ptr()
        self.net = net

    def _report_fallbacks(self):
        if self._fb_event is not None and self._fb_event.query():
            n = int(self._fb_host[0])
            if n > self._fb_seen:
                import warnings
                warnings.warn("fnssl: %d LSTM layer launch(es) so far could not keep every member workgroup of a cluster-resident "
                              "kernel resident (shared, partitioned or busy device?) and were recomputed by the per-wave kernels: "
                              "results are unaffected, throughput is" % n, RuntimeWarning, stacklevel=3)
                self._fb_seen = n
            self._fb_event = None

    def _poll_fallbacks(self):
        if self._fb_host is not None and self._fb_event is None:
            self._fb_host.copy_(self.fallbacks, non_blocking=True)
            self._fb_event = torch.cuda.Event()
            self._fb_event.record(torch.cuda.current_stream(self.device))

    def _dev(self, a):
        t = a.detach() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a, dtype=np.float32))
        return t.to(self.device, torch.float32).contiguous()

    def _run(self, x0, out, chunk_pairs):
        nb, nt, nf, _ = x0.shape
        lib = _lib.load()
        ws = _workspace(lib.fnssl_forward_workspace_bytes(nb, nf, nt, int(self.is_online), chunk_pairs), x0.device, "forward")
        check(lib.fnssl_forward(C.byref(self.net), _ptr(x0), nb, nf, nt, _ptr(out), _ptr(ws), ws.numel(),
                                chunk_pairs, _stream()), "forward")

    @on_device
    def forward(self, x0: torch.Tensor, chunk_pairs: int = 0) -> torch.Tensor:
        """x0 [nb', nt, nf, input_size] -> [nb', nt//12, 2*nf] (or [.., 180] with the DOA layer)."""
        _need_dev(x0)
        x0 = x0.contiguous()
        nb, nt, nf, cin = x0.shape
        if cin != self.input_size:
            raise RuntimeError("fnssl.forward: expected %d input channels, got %d" % (self.input_size, cin))
        last = 180 if self.is_doa else 2 * nf
        out = torch.empty((nb, nt // SEG_FRAMES, last), dtype=torch.float32, device=x0.device)
        self._report_fallbacks()
        self._run(x0, out, chunk_pairs)
        self._poll_fallbacks()
        return out


# --------------------------------------------------------------------------- #
# measurement hooks
# --------------------------------------------------------------------------- #
@on_device
def occupy_cus(nblocks: int, stop, max_ms: int = 2000, lds_bytes: int = 160 * 1024, stream=None):
    """Diagnostic (fnssl_occupy_cus): ``nblocks`` workgroups, each holding ``lds_bytes`` of LDS (default: a whole CU), idle on
    ``stream`` (default: the current one) until ``stop[0]`` is non-zero or ``max_ms`` pass — what RCCL's persistent all-reduce
    kernels do to the CUs under an overlapped backward.  ``stop``: int32 tensor of 1 + nblocks words in pinned host (or
    device) memory, zeroed by the caller; workgroup b sets ``stop[1 + b]`` when it has become resident."""
    if stop.dtype != torch.int32 or stop.numel() < 1 + int(nblocks):
        raise RuntimeError("fnssl.occupy_cus: stop must hold 1 + nblocks int32 words")
    s = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(_lib.load().fnssl_occupy_cus(int(nblocks), int(lds_bytes), C.c_void_p(stop.data_ptr()), int(max_ms), s), "occupy_cus")



def timing_enable(on: bool):
    check(_lib.load().fnssl_timing_enable(1 if on else 0), "timing_enable")


def timing_select(name=None):
    """Bracket only launches of kernel ``name`` (None: all)."""
    check(_lib.load().fnssl_timing_select(name.encode() if name else None), "timing_select")


def timing_collect(cap: int = 64):
    names = ((C.c_char * 64) * cap)()
    ms = (C.c_double * cap)()
    cnt = (C.c_longlong * cap)()
    fl = (C.c_double * cap)()
    n = _lib.load().fnssl_timing_collect(cap, names, ms, cnt, fl)
    return {names[i].value.decode(): {"ms": ms[i], "count": cnt[i], "flops": fl[i]} for i in range(n)}
