"""The Lightning predict entry of the reference (FN-SSL/Lightning/main.py:81-134,184-225),
restated over the HIP path: ``MyModel.predict_step(batch[nb, nch, ns], batch_idx)``
returns the raw DP-IPD predictions ``[nb*np, nt//12, 512]``.

``pytorch_lightning`` is optional: with it installed ``MyModel`` is a
``LightningModule`` (so ``Trainer.predict`` / ``LightningCLI`` drive it exactly
like the reference's ``main.py predict``); without it the same class derives
from ``nn.Module`` and ``predict_step`` is called directly (``Predict.py``).
Training/validation steps, the numpy DP-IPD target generator and the DOA
metrics are outside this path.
"""
import torch

import Model as at_model
from fnssl import ops

try:  # optional, absent in the build image
    from pytorch_lightning import LightningModule as _Base
except Exception:  # pragma: no cover
    _Base = torch.nn.Module


class MyModel(_Base):
    def __init__(self, tar_useVAD: bool = True, ch_mode: str = 'MM', res_the: int = 37, res_phi: int = 73,
                 fs: int = 16000, win_len: int = 512, nfft: int = 512, win_shift_ratio: float = 0.5,
                 method_mode: str = 'IDL', source_num_mode: str = 'KNum', max_num_sources: int = 1,
                 return_metric: bool = True, exp_name: str = 'exp', compile: bool = False,
                 device: str = "cuda"):
        super().__init__()
        if (win_len, nfft, win_shift_ratio) != (512, 512, 0.5):
            raise ValueError("the MI355X path is built for win_len = nfft = 512, hop 256 (main.py:38-44)")
        self.arch = at_model.FN_SSL()
        self.ch_mode = ch_mode
        self.nfft = nfft
        self.dev = device
        self.fre_range_used = range(1, int(self.nfft / 2) + 1, 1)
        self.eval()

    def forward(self, x):
        return self.arch(x)

    def data_preprocess(self, mic_sig_batch=None, gt_batch=None, vad_batch=None, eps=1e-6, nor_flag=True):
        """Input half of main.py:200-225: [nb, ns, nch] -> [[nb*np, 4, 256, nt]] (reference layout)."""
        if gt_batch is not None:
            raise NotImplementedError("ground-truth DP-IPD targets (numpy DPIPD, main.py:227-265) are outside "
                                      "the forward path")
        if not nor_flag:
            raise NotImplementedError("nor_flag=False is not part of the path")
        data = []
        if mic_sig_batch is not None:
            mic_sig_batch = mic_sig_batch.to(self.dev)
            data += [ops.preprocess(mic_sig_batch, self.ch_mode, eps, layout=1)]
        return data

    @torch.no_grad()
    def predict_step(self, batch, batch_idx: int = 0):
        """batch [nb, nch, ns] -> preds [nb*np, nt//12, 512]  (main.py:184-189).

        Fused: STFT -> pair features land directly in the [nb', nt, nf, 4] layout the
        LSTM kernels read, so neither the reference's [nb', 4, nf, nt] tensor nor its
        permute is materialised.
        """
        sig = batch.permute(0, 2, 1).to(self.dev)
        x0 = ops.preprocess(sig, self.ch_mode, 1e-6, layout=0)
        return self.arch.forward_seq(x0)
