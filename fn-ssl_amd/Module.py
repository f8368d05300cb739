"""Drop-ins for the hot-path classes of the reference's ``Module.py``
(FN-SSL/Module.py:28-68 STFT, :376-404 AddChToBatch, :406-421 RemoveChFromBatch).

The signal front end and — as the first "next" row of SURVEY.md §8f — the IPD->DOA back end
(DPIPD templates :424-519, SourceDetectLocalize 'IDL' :516-577, PredDOA.predgt2DOA :690-727)
live here; metrics and plotting of the reference's Module.py are out of scope.  The
fused front end used by ``predict_step`` is ``fnssl.ops.preprocess`` (one STFT
kernel + one scan + one pack kernel); the classes below exist so code written
against the reference's per-stage API keeps working, with the same shapes and
dtypes.
"""
import numpy as np
import torch
import torch.nn as nn

from fnssl import doa as fdoa
from fnssl import ops


class STFT(nn.Module):
    """signal [nb, ns, nch] -> complex64 [nb, nf=257, nt, nch]  (Hann-512, hop 256, center=False)."""

    def __init__(self, win_len, win_shift_ratio, nfft, win='hann'):
        super(STFT, self).__init__()
        if win_len != 512 or nfft != 512 or win_shift_ratio != 0.5 or win != 'hann':
            raise ValueError("STFT: the MI355X path is built for the reference's constants "
                             "(win_len=nfft=512, win_shift_ratio=0.5, hann; Predict.py:33-35)")
        self.win_len = win_len
        self.win_shift_ratio = win_shift_ratio
        self.nfft = nfft
        self.win = win

    def forward(self, signal):
        spec, _ = ops.stft(signal)                       # [nb, nch, nt, 257, 2]
        return torch.view_as_complex(spec).permute(0, 3, 2, 1)


class AddChToBatch(nn.Module):
    """[nb, nch, ...] -> [nb*np, 2, ...]: mic pairs (0, j) ('M') or (i<j) ('MM'), reference row order.

    Pure data movement (an index gather on device memory); the fused feature
    kernel does this implicitly and never materialises the re-batched spectrum.
    """

    def __init__(self, ch_mode):
        super(AddChToBatch, self).__init__()
        if ch_mode not in ('M', 'MM'):
            raise ValueError("ch_mode must be 'M' or 'MM'")
        self.ch_mode = ch_mode

    def forward(self, data):
        nb, nch = data.shape[0], data.shape[1]
        if self.ch_mode == 'M':
            pairs = [(0, j) for j in range(1, nch)]
        else:
            pairs = [(i, j) for i in range(nch - 1) for j in range(i + 1, nch)]
        idx = torch.tensor(pairs, dtype=torch.long, device=data.device)      # [np, 2]
        out = data[:, idx]                                                   # [nb, np, 2, ...]
        return out.reshape((nb * len(pairs), 2) + tuple(data.shape[2:])).contiguous()


class RemoveChFromBatch(nn.Module):
    """[nb*nmic, ...] -> [nb, nmic, ...]"""

    def __init__(self, ch_mode):
        super(RemoveChFromBatch, self).__init__()
        self.ch_mode = ch_mode

    def forward(self, data, nb):
        nmic = int(data.shape[0] / nb)
        return data.reshape((nb, nmic) + tuple(data.shape[1:])).contiguous()


class DPIPD(nn.Module):
    """DP-IPD template bank exp(-j 2 pi f tau) for a grid of candidate directions (host, numpy;
    built once per array geometry).  ``forward(source_doa)`` returns (template, dpipd, doa_candidate) like the
    reference (Module.py:464-498; numpy in, numpy out); the training path's targets come from one HIP kernel
    (``fnssl.doa.dpipd_targets``, used by ``predict_step.MyModel.data_preprocess``)."""

    def __init__(self, ndoa_candidate, mic_location, nf=257, fre_max=8000, ch_mode='M', speed=343.0):
        super(DPIPD, self).__init__()
        self.ndoa_candidate = ndoa_candidate
        self.mic_location = mic_location
        self.nf, self.fre_max, self.speed, self.ch_mode = nf, fre_max, speed, ch_mode
        self.dpipd_template, self.doa_candidate = fdoa.dpipd_templates(
            mic_location, ndoa_candidate[0], ndoa_candidate[1], nf, fre_max, ch_mode, speed)

    def forward(self, source_doa=None):
        dpipd = None
        if source_doa is not None:
            dpipd = fdoa.dpipd_of_sources(source_doa, self.mic_location, self.nf, self.fre_max, self.ch_mode, self.speed)
        return self.dpipd_template, dpipd, self.doa_candidate


class SourceDetectLocalize(nn.Module):
    """Localisation and voice-activity detection on device: iterative ('IDL', Module.py:525-577) or by peak detection over
    the spatial spectrum ('PD', Module.py:580-622)."""

    def __init__(self, max_num_sources, source_num_mode='kNum', meth_mode='IDL'):
        super(SourceDetectLocalize, self).__init__()
        self.max_num_sources = max_num_sources
        self.source_num_mode = source_num_mode
        self.meth_mode = meth_mode

    def forward(self, pred_ipd, dpipd_template, doa_candidate):
        """pred_ipd [nb, nt, 2nf, np], dpipd_template [nele, nazi, 2nf, np] ->
        (DOAs [nb, nt, 2, ns], VADs [nb, nt, ns], spatial spectrum [nb, nt, nele, nazi])"""
        nb = pred_ipd.shape[0]
        nele, nazi = dpipd_template.shape[:2]
        ele = torch.as_tensor(np.asarray(doa_candidate[0])).to(pred_ipd.device, torch.float32)
        azi = torch.as_tensor(np.asarray(doa_candidate[1])).to(pred_ipd.device, torch.float32)
        if self.meth_mode == 'PD':
            return self._peak_detection(pred_ipd, dpipd_template, ele, azi)
        if self.meth_mode != 'IDL':
            raise Exception('Localizion method is unrecognized')                 # Module.py:621
        idx, vads, ss = fdoa.localize(pred_ipd.detach(), dpipd_template.to(pred_ipd.device), nb,
                                      self.max_num_sources, self.source_num_mode)
        idx = idx.long()
        doas = torch.stack((ele[idx // nazi], azi[idx % nazi]), dim=2)       # [nb, nt, 2, ns]
        return doas, vads, ss

    def _peak_detection(self, pred_ipd, dpipd_template, ele, azi):
        """'PD' as the reference RUNS it (measured on the real reference, tests/golden/make_golden_pd.py): the slice assignment
        at Module.py:615 only goes through for max_num_sources = 2, and it lands as DOAs[b, t, SOURCE, (ele, azi)] — the
        transpose of the 'IDL' layout; a frame with one peak gives both sources that peak; any other source count, or a
        frame without a peak, raises RuntimeError there and here."""
        ns = int(self.max_num_sources)
        if ns != 2:
            raise RuntimeError("SourceDetectLocalize('PD'): the reference's assignment pred_DOAs[b, t, :, :] = pred_DOA.transpose(1, 0) "
                               "(Module.py:615) only accepts max_num_sources = 2, got %d" % ns)
        nb = pred_ipd.shape[0]
        nazi = dpipd_template.shape[1]
        idx, val, cnt, ss = fdoa.localize_pd(pred_ipd.detach(), dpipd_template.to(pred_ipd.device), nb, ns, self.source_num_mode)
        if int(cnt.min()) == 0:                                                  # (one 4-byte read-back: 'PD' is not a hot path)
            raise RuntimeError("SourceDetectLocalize('PD'): a frame of the spatial spectrum has no peak (Module.py:615 raises)")
        one = (cnt == 1).unsqueeze(-1)
        idx = torch.where(one, idx[..., :1].expand_as(idx), idx).long()          # a single peak broadcasts to both sources
        val = torch.where(one, val[..., :1].expand_as(val), val)
        doas = torch.stack((ele[idx // nazi], azi[idx % nazi]), dim=3)           # [nb, nt, source, (ele, azi)]
        vads = torch.ones_like(val) if self.source_num_mode == 'kNum' else val
        if self.source_num_mode not in ('kNum', 'unkNum'):
            vads = torch.zeros_like(val)                                         # neither branch of :616-619 writes
        return doas, vads, ss


class PredDOA(nn.Module):
    """DP-IPD predictions -> DOA tracks (prediction half of the reference's PredDOA, Module.py:650-727).
    ``mic_location`` defaults to the reference's hard-coded two-microphone array."""

    def __init__(self, method_mode='IDL', source_num_mode='kNum', cuda_activated=True, max_num_sources=1,
                 res_the=37, res_phi=73, fs=16000, nfft=512, ch_mode='MM', device="cuda", mic_location=None):
        super(PredDOA, self).__init__()
        self.nfft = nfft
        self.fre_max = fs / 2
        self.ch_mode = ch_mode
        self.dev = device
        if mic_location is None:
            mic_location = np.array(((-0.04, 0.0, 0.0), (0.04, 0.0, 0.0)))
        self.removebatch = RemoveChFromBatch(ch_mode=self.ch_mode)
        self.gerdpipd = DPIPD(ndoa_candidate=[res_the, res_phi], mic_location=mic_location,
                              nf=int(self.nfft / 2) + 1, fre_max=self.fre_max, ch_mode=self.ch_mode, speed=340)
        self.sourcelocalize = SourceDetectLocalize(max_num_sources=int(max_num_sources),
                                                   source_num_mode=source_num_mode, meth_mode=method_mode)
        bank, self.doa_candidate = fdoa.template_bank(self.gerdpipd.dpipd_template)
        self.register_buffer("bank", torch.from_numpy(bank), persistent=False)

    def predgt2DOA(self, pred_batch=None, gt_batch=None, time_pool_size=None):
        if pred_batch is not None:
            pred_ipd = pred_batch.detach()
            if time_pool_size is not None:       # mean over blocks of time_pool_size segments, floor (Module.py:723-730)
                from fnssl import ops as _ops
                nbp, nt, nf2 = pred_ipd.shape
                pred_ipd = _ops.avgpool_time(pred_ipd.float().reshape(nbp, 1, nt, nf2), int(time_pool_size)).reshape(nbp, -1, nf2)
            npair = self.bank.shape[-1]
            nb = pred_ipd.shape[0] // npair
            idx, vads, ss = fdoa.localize(pred_ipd, self.bank.to(pred_ipd.device), nb,
                                          self.sourcelocalize.max_num_sources, self.sourcelocalize.source_num_mode)
            nazi = self.bank.shape[1]
            ele = torch.as_tensor(self.doa_candidate[0]).to(pred_ipd.device, torch.float32)
            azi = torch.as_tensor(self.doa_candidate[1]).to(pred_ipd.device, torch.float32)
            idx = idx.long()
            pred_batch = {'doa': torch.stack((ele[idx // nazi], azi[idx % nazi]), dim=2),
                          'vad_sources': vads, 'spatial_spectrum': ss}
        if gt_batch is not None:
            for key in gt_batch.keys():
                gt_batch[key] = gt_batch[key].detach()
        return pred_batch, gt_batch
