#!/usr/bin/env python3
"""Golden vectors for IPDnet2's waveform front end (G15), generated from the REAL reference in the build container:
/root/reference/IPDnet2/Module.py::STFT (nfft 512, hop int(512 * 0.625) = 320, center=True; :47-64) and
/root/reference/IPDnet2/utils_.py::forgetting_norm, composed exactly as IPDnet2/run_IPDnet2.py:277-288 composes them
(data_preprocess itself also needs a simulated scene and an array geometry).  Separate script: the IPDnet2 tree ships
its own `Module` / `utils_` modules.  Data only: seeds, shapes, reference outputs."""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/IPDnet2")
for m in ("soundfile", "webrtcvad"):
    sys.modules.setdefault(m, types.ModuleType(m))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import Module as at_module  # noqa: E402  (reference IPDnet2/Module.py)
from utils_ import forgetting_norm  # noqa: E402  (reference IPDnet2/utils_.py)

torch.set_num_threads(8)


def rs_randn(seed, shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(np.float32)


@torch.no_grad()
def main():
    arrs = {}
    dostft = at_module.STFT(win_len=512, win_shift_ratio=0.625, nfft=512)          # run_IPDnet2.py:91-93,135-136
    fre_range_used = range(1, 257)                                                  # :139
    # (seed, nb, ns, nch, sample_length): ns not a multiple of the hop, ns a multiple of it, a signal shorter than one
    # window (every frame touches a reflected edge), and a short sample_length so both branches of forgetting_norm run
    cases = [(1800, 2, 320 * 11 + 77, 5, 249), (1801, 1, 320 * 8, 3, 249), (1802, 1, 400, 2, 249),
             (1803, 2, 320 * 14 + 5, 4, 6)]
    for ci, (seed, nb, ns, nch, sl) in enumerate(cases):
        sig = rs_randn(seed, (nb, ns, nch), 0.1)
        stft = dostft(signal=torch.from_numpy(sig))                                 # :277  [nb, 257, nt, nch]
        reb = stft.permute(0, 3, 1, 2)                                              # :279
        mean_value = forgetting_norm(torch.abs(reb), sample_length=sl)              # :282-283
        feat = torch.cat((torch.real(reb) / (mean_value + 1e-6), torch.imag(reb) / (mean_value + 1e-6)), dim=1)
        arrs["c%d_cfg" % ci] = np.array([seed, nb, ns, nch, sl])
        arrs["c%d_feat" % ci] = feat[:, :, fre_range_used, :].numpy()              # :288
        if ci == 0:
            arrs["c0_stft"] = stft.numpy()
            arrs["c0_mu"] = mean_value.numpy()
    path = os.path.join(HERE, "g15_ipdnet2_frontend.npz")
    np.savez(path, **arrs)
    print("g15_ipdnet2_frontend %.1f KB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
