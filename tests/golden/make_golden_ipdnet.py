#!/usr/bin/env python3
"""Golden vectors for the IPDnet row (G10), generated from the REAL reference
/root/reference/IPDnet/FixedAarryIPDnet.py in the build container (separate script because the
IPDnet tree ships its own `Module` / `utils_` modules, which would clash with FN-SSL's).
Data only: seeds, shapes and reference outputs."""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
sys.path.insert(0, "/root/reference/IPDnet")
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
sys.modules.setdefault("webrtcvad", types.ModuleType("webrtcvad"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import FixedAarryIPDnet as ref  # noqa: E402  (reference)
from fnssl import weights as W  # noqa: E402

torch.set_num_threads(8)


def rs_randn(seed, shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(np.float32)


@torch.no_grad()
def main():
    arrs = {}
    cases = [  # (input_size, hidden, max_track, is_online, x shape)
        (4, 128, 2, True, (2, 4, 16, 24)),
        (16, 256, 2, True, (1, 16, 32, 24)),          # 8-mic
        (4, 128, 2, False, (2, 4, 16, 29)),
        (4, 128, 2, True, (1, 4, 256, 12)),
    ]
    for ci, (isz, hid, mt, online, shape) in enumerate(cases):
        sd = W.make_ipdnet_state(1500 + ci, isz, hid, mt, online)
        net = ref.IPDnet(input_size=isz, hidden_size=hid, max_track=mt, is_online=online).eval()
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        x = rs_randn(1600 + ci, shape)
        y = net(torch.from_numpy(x))
        arrs["c%d_cfg" % ci] = np.array([isz, hid, mt, int(online), 1500 + ci, 1600 + ci] + list(shape))
        arrs["c%d_out" % ci] = y.numpy()
    # chunk-wise offline inference (offline_inference=True, :96-100, :114-116): 40 frames in segments of 24
    sd = W.make_ipdnet_state(1520, 4, 128, 2, False)
    net = ref.IPDnet(input_size=4, hidden_size=128, max_track=2, is_online=False, n_seg=24).eval()
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    arrs["seg_out"] = net(torch.from_numpy(rs_randn(1620, (2, 4, 16, 40))), offline_inference=True).numpy()
    # input features: the reference's own STFT module and forgetting_norm composed exactly as
    # runIPDnetOn.py:240-254 does (data_preprocess itself also needs a simulated acoustic scene)
    import Module as at_module  # noqa: E402  (reference IPDnet/Module.py)
    from utils_ import forgetting_norm  # noqa: E402  (reference)
    sig = rs_randn(1630, (2, 256 * 14, 4), 0.1)
    stft = at_module.STFT(win_len=512, win_shift_ratio=0.5, nfft=512)(signal=torch.from_numpy(sig))
    reb = stft.permute(0, 3, 1, 2)
    mean_value = forgetting_norm(torch.abs(reb), sample_length=280)
    feat = torch.cat((torch.real(reb) / (mean_value + 1e-6), torch.imag(reb) / (mean_value + 1e-6)), dim=1)
    arrs["feat_out"] = feat[:, :, range(1, 257), :].numpy()
    # the causal conv block alone
    blk = ref.CausCnnBlock(inp_dim=20, out_dim=6).eval()
    sdc = {"conv%d.weight" % (i + 1): rs_randn(1700 + i, s, 0.1) for i, s in
           enumerate([(128, 20, 3, 3), (128, 128, 3, 3), (6, 128, 3, 3)])}
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in sdc.items()})
    xc = rs_randn(1710, (2, 20, 7, 26))
    arrs["cnn_out"] = blk(torch.from_numpy(xc)).numpy()
    path = os.path.join(HERE, "g10_ipdnet.npz")
    np.savez(path, **arrs)
    print("g10_ipdnet %.1f KB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
