"""IPD -> DOA back end: host-side DP-IPD template generator (numpy, built once per array
geometry) and the device localisation op.

Reference: ``DPIPD`` (FN-SSL/Lightning/Module.py:424-519), the bank selection inside
``PredDOA.predgt2DOA`` (:702-716) and ``SourceDetectLocalize`` 'IDL' (:525-577).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _need_dev, _ptr, _stream, on_device


def pair_list(nmic: int, ch_mode: str):
    if ch_mode == "M":
        return [(0, j) for j in range(1, nmic)]
    if ch_mode == "MM":
        return [(i, j) for i in range(nmic - 1) for j in range(i + 1, nmic)]
    raise ValueError("Microphone channel mode unrecognised")


def dpipd_templates(mic_location, nele: int = 37, nazi: int = 73, nf: int = 257, fre_max: float = 8000.0,
                    ch_mode: str = "MM", speed: float = 340.0):
    """exp(-j 2 pi f tau) for every candidate direction and mic pair (DPIPD.__init__, Module.py:429-463).

    Returns (complex64 [nele, nazi, nf, np], [ele_candidate, azi_candidate]).
    """
    mic = np.asarray(mic_location, dtype=np.float64)
    ele = np.linspace(0, np.pi, nele)
    azi = np.linspace(-np.pi, np.pi, nazi)
    fre = np.linspace(0.0, fre_max, nf)
    unit = np.stack([np.outer(np.sin(ele), np.cos(azi)), np.outer(np.sin(ele), np.sin(azi)),
                     np.tile(np.cos(ele), [nazi, 1]).transpose()], axis=2)
    pairs = pair_list(mic.shape[-2], ch_mode)
    out = np.empty((nele, nazi, nf, len(pairs)), dtype=np.complex64)
    for p, (i, j) in enumerate(pairs):
        itd = np.dot(unit, mic[j] - mic[i]) / speed
        out[..., p] = np.exp(1j * (-2 * np.pi * fre[None, None, :] * itd[:, :, None]))
    return out, [ele, azi]


def dpipd_of_sources(source_doa, mic_location, nf: int = 257, fre_max: float = 8000.0, ch_mode: str = "MM",
                     speed: float = 340.0) -> np.ndarray:
    """Host form of ``DPIPD.forward(source_doa)`` (Module.py:464-498): source_doa [nb, ntime, 2, nsource] (numpy, radians)
    -> complex64 [nb, ntime, nf, np, nsource] = exp(+j 2 pi f tau), tau = r(doa) . (mic_i - mic_j) / speed for pair
    (i, j).  Kept for API compatibility of the drop-in class; the training path uses ``dpipd_targets`` (one HIP kernel)."""
    doa = np.asarray(source_doa, dtype=np.float64).transpose(0, 1, 3, 2)           # [nb, ntime, nsource, 2]
    mic = np.asarray(mic_location, dtype=np.float64)
    fre = np.linspace(0.0, fre_max, nf)
    r = np.stack([np.sin(doa[..., 0]) * np.cos(doa[..., 1]), np.sin(doa[..., 0]) * np.sin(doa[..., 1]), np.cos(doa[..., 0])],
                 axis=3)                                                             # [nb, ntime, nsource, 3]
    pairs = pair_list(mic.shape[-2], ch_mode)
    out = np.empty(doa.shape[:3] + (nf, len(pairs)), dtype=np.complex64)
    for p, (i, j) in enumerate(pairs):
        itd = np.dot(r, mic[i] - mic[j]) / speed                                     # ITD[m1 = i, m2 = j] (:488)
        out[..., p] = np.exp(1j * (2 * np.pi * fre[None, None, None, :] * itd[..., None]))
    return out.transpose(0, 1, 3, 4, 2)                                              # [nb, ntime, nf, np, nsource]


@on_device
def dpipd_targets(doa: torch.Tensor, vad, mic_location, ch_mode: str = "MM", bin0: int = 1, nf_used: int = 256,
                  nbins: int = 257, fre_max: float = 8000.0, speed: float = 340.0, use_vad: bool = True):
    """The training targets on device (``fnssl_dpipd_targets``; reference main.py:227-262): doa [nb, nseg, 2, ns],
    vad [nb, nseg, nvad, ns] (or None) -> (ipd [nb, nseg, 2 * nf_used, np], vad_mean [nb, nseg, ns])."""
    _need_dev(doa, vad)
    doa = doa.contiguous()
    nb, nseg, two, ns = doa.shape
    if two != 2:
        raise RuntimeError("fnssl.doa.dpipd_targets: doa must be [nb, nseg, 2, nsource]")
    if vad is not None:
        vad = vad.contiguous()
        if vad.ndim != 4 or vad.shape[0] != nb or vad.shape[1] != nseg or vad.shape[3] != ns:
            raise RuntimeError("fnssl.doa.dpipd_targets: vad %s does not match doa %s" % (tuple(vad.shape), tuple(doa.shape)))
    mic = torch.as_tensor(np.asarray(mic_location, dtype=np.float32).reshape(-1, 3)).to(doa.device).contiguous()
    nmic = mic.shape[0]
    npair = len(pair_list(nmic, ch_mode))
    ipd = torch.empty((nb, nseg, 2 * nf_used, npair), dtype=torch.float32, device=doa.device)
    vmean = torch.empty((nb, nseg, ns), dtype=torch.float32, device=doa.device)
    _lib.check(_lib.load().fnssl_dpipd_targets(_ptr(doa), _ptr(vad), nb, nseg, 0 if vad is None else vad.shape[2], ns, _ptr(mic), nmic,
                                               _lib.CH_MODE[ch_mode], bin0, nf_used, nbins, float(fre_max), float(speed),
                                               1 if use_vad else 0, _ptr(ipd), _ptr(vmean), _stream()), "dpipd_targets")
    return ipd, vmean


def template_bank(template: np.ndarray):
    """The bank PredDOA searches (Module.py:702-716): [cos | sin] of bins 1..256, middle elevation
    row, upper azimuth half; candidates ele = pi/2, azi = linspace(0, pi, 37)."""
    nele, nazi = template.shape[:2]
    t = np.concatenate((template.real[:, :, 1:257, :], template.imag[:, :, 1:257, :]), axis=2).astype(np.float32)
    t = t[int((nele - 1) / 2):int((nele - 1) / 2) + 1, int((nazi - 1) / 2):nazi, :, :]
    cand = [np.linspace(np.pi / 2, np.pi / 2, 1), np.linspace(0, np.pi, 37)]
    return np.ascontiguousarray(t), cand


@on_device
def localize(pred: torch.Tensor, bank: torch.Tensor, nb: int, max_num_sources: int = 1,
             source_num_mode: str = "kNum"):
    """Iterative detection/localisation on device.

    pred: the network output [nb*np, nt, 2nf], or the reference's re-batched [nb, nt, 2nf, np] tensor
    (any strides — it is read in place); bank [nele, nazi, 2nf, np] device tensor.
    Returns (idx int32 [nb, nt, ns] into the flattened (ele, azi) grid, vad [nb, nt, ns],
    ss [nb, nt, nele, nazi]).
    """
    _need_dev(pred, bank)
    if source_num_mode not in ("kNum", "unkNum"):
        raise RuntimeError("source_num_mode must be 'kNum' or 'unkNum'")
    bank = bank.contiguous()
    nele, nazi, nf2, np_ = bank.shape
    if pred.ndim == 3:
        nbp, nt, nf2p = pred.shape
        if nbp != nb * np_:
            raise RuntimeError("fnssl.doa.localize: %d rows != nb*np = %d*%d" % (nbp, nb, np_))
        s0, st, sk = pred.stride()
        sb, sp = s0 * np_, s0
    elif pred.ndim == 4:
        nbb, nt, nf2p, npp = pred.shape
        if nbb != nb or npp != np_:
            raise RuntimeError("fnssl.doa.localize: pred %s does not match nb=%d, np=%d" % (tuple(pred.shape), nb, np_))
        sb, st, sk, sp = pred.stride()
    else:
        raise RuntimeError("fnssl.doa.localize: pred must be 3-D or 4-D")
    if nf2p != nf2:
        raise RuntimeError("fnssl.doa.localize: pred has %d frequency features, bank %d" % (nf2p, nf2))
    ns = int(max_num_sources)
    ss = torch.empty((nb, nt, nele, nazi), dtype=torch.float32, device=pred.device)
    idx = torch.empty((nb, nt, ns), dtype=torch.int32, device=pred.device)
    vad = torch.empty((nb, nt, ns), dtype=torch.float32, device=pred.device)
    _lib.check(_lib.load().fnssl_ipd2doa(_ptr(pred), sb, sp, st, sk, _ptr(bank), nb, np_, nt, nf2, nele * nazi, ns,
                                         1 if source_num_mode == "unkNum" else 0, _ptr(ss),
                                         C.c_void_p(idx.data_ptr()), _ptr(vad), _stream()), "ipd2doa")
    return idx, vad, ss


@on_device
def localize_pd(pred: torch.Tensor, bank: torch.Tensor, nb: int, max_num_sources: int = 2,
                source_num_mode: str = "kNum"):
    """Peak-detection localisation on device (SourceDetectLocalize, meth_mode 'PD', Module.py:580-622): the spatial
    spectrum (fnssl_ipd2doa) and the ``max_num_sources`` largest local maxima of every frame (fnssl_doa_peaks).

    Returns (idx int32 [nb, nt, ns] flat (ele, azi) cell or -1, val [nb, nt, ns] the peaks' spectrum values,
    count int32 [nb, nt] = min(peaks, ns), ss [nb, nt, nele, nazi])."""
    ns = int(max_num_sources)
    _idx, _vad, ss = localize(pred, bank, nb, 1, "kNum")         # the spectrum of the first pass (the argmax is not used)
    nele, nazi = bank.shape[:2]
    nt = ss.shape[1]
    idx = torch.empty((nb, nt, ns), dtype=torch.int32, device=pred.device)
    val = torch.empty((nb, nt, ns), dtype=torch.float32, device=pred.device)
    cnt = torch.empty((nb, nt), dtype=torch.int32, device=pred.device)
    _lib.check(_lib.load().fnssl_doa_peaks(_ptr(ss), nb * nt, nele, nazi, ns, C.c_void_p(idx.data_ptr()), _ptr(val),
                                           C.c_void_p(cnt.data_ptr()), _stream()), "doa_peaks")
    return idx, val, cnt, ss
