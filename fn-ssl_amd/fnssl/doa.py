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
