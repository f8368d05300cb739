"""Drop-ins for the hot-path classes of the reference's ``Module.py``
(FN-SSL/Module.py:28-68 STFT, :376-404 AddChToBatch, :406-421 RemoveChFromBatch).

Only the signal front end lives here; the IPD->DOA back end, metrics and
plotting of the reference's Module.py are out of scope.  The
fused front end used by ``predict_step`` is ``fnssl.ops.preprocess`` (one STFT
kernel + one scan + one pack kernel); the classes below exist so code written
against the reference's per-stage API keeps working, with the same shapes and
dtypes.
"""
import torch
import torch.nn as nn

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
