"""Drop-in for the hot-path class of the reference's ``IPDnet2/Module.py``: ``STFT`` (:28-64) — the IPDnet2 tree's
transform is CENTRED (``torch.stft(..., center=True)``, reflect padding) and its run script uses
``win_shift_ratio = 0.625`` (hop 320; IPDnet2/run_IPDnet2.py:91-93,135-136), unlike FN-SSL's ``Module.STFT``.
The fused front end the network entry uses is ``fnssl.ops.preprocess_ipdnet2`` (STFT kernel + one scan + one pack
kernel = run_IPDnet2.py:277-288); this class keeps code written against the per-stage API working.  Metrics,
plotting and target generation of the reference's Module.py are out of scope."""
import torch
import torch.nn as nn

from fnssl import ops


class STFT(nn.Module):
    """signal [nb, ns, nch] -> complex64 [nb, 257, nt = ns // hop + 1, nch]  (Hann-512, centred, reflect-padded)."""

    def __init__(self, win_len, win_shift_ratio, nfft, win='hann'):
        super(STFT, self).__init__()
        if win_len != 512 or nfft != 512 or win != 'hann':
            raise ValueError("STFT: the MI355X path is built for win_len = nfft = 512, hann (run_IPDnet2.py:91-93)")
        self.win_len = win_len
        self.win_shift_ratio = win_shift_ratio
        self.nfft = nfft
        self.win = win

    def forward(self, signal):
        hop = int(self.win_len * self.win_shift_ratio)                 # Module.py:51
        spec, _ = ops.stft(signal, hop=hop, center=True)               # [nb, nch, nt, 257, 2]
        return torch.view_as_complex(spec).permute(0, 3, 2, 1)
