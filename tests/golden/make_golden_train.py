#!/usr/bin/env python3
"""Golden vectors for the training step (G13), generated from the REAL reference
(/root/reference/FN-SSL/Lightning/Model.py FN_SSL in train() mode, autograd backward, torch.optim.Adam as
configured at main.py:269-271) in the build container.  The only substitution: each nn.Dropout samples
nothing — its forward is replaced by a multiplication with the deterministic keep-scale tensor that the
HIP path uses (oracle.train_ref.dropout_scale), because torch's Bernoulli stream cannot be reproduced
elsewhere.  cal_loss (main.py:191-198) is restated with the reference's RemoveChFromBatch.

Data only: the 2.5 M gradients are stored as per-tensor projections (L2 norm, dot product with a seeded
random tensor, the first 16 entries) plus the loss, the prediction and the same projections of the
Adam-updated parameters."""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/FN-SSL/Lightning")
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import Model as ref_model  # noqa: E402  (reference)
import Module as ref_module  # noqa: E402  (reference)
from fnssl import weights as W  # noqa: E402
from oracle import train_ref as T  # noqa: E402

torch.set_num_threads(8)


def rs_randn(seed, shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(np.float32)


def project(name_index, a):
    a = np.asarray(a, dtype=np.float64)
    r = rs_randn(5000 + name_index, a.shape).astype(np.float64)
    return np.array([np.sqrt((a * a).sum()), (a * r).sum()]), a.reshape(-1)[:16].astype(np.float32)


def main():
    arrs = {}
    for ci, (online, nb, npair, nf, nt, seed) in enumerate([(True, 1, 2, 8, 24, 11), (False, 2, 1, 5, 13, 12)]):
        sd = W.make_fnssl_state(1800 + ci, 4, 256, online)
        net = ref_model.FN_SSL(is_online=online)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        net.train()
        nbp = nb * npair
        masks = T.make_masks(seed, nbp, nt, nf, 256, online)
        for k, blk in enumerate((net.block_1, net.block_2, net.block_3)):
            mf = masks[2 * k].reshape(nbp * nt, nf, -1)
            mn = masks[2 * k + 1].permute(0, 2, 1, 3).reshape(nbp * nf, nt, -1)
            blk.dropout_full.forward = (lambda x, m=mf: x * m)
            blk.dropout_narr.forward = (lambda x, m=mn: x * m)
        x = rs_randn(1810 + ci, (nbp, 4, nf, nt))
        gt = rs_randn(1820 + ci, (nb, nt // 12, 2 * nf, npair), 0.5)
        pred = net(torch.from_numpy(x))
        # cal_loss, main.py:191-198
        removebatch = ref_module.RemoveChFromBatch(ch_mode="MM")
        reb = removebatch(pred, nb).permute(0, 2, 3, 1)
        loss = torch.nn.functional.mse_loss(reb.contiguous(), torch.from_numpy(gt).contiguous())
        opt = torch.optim.Adam(net.parameters(), lr=0.001)     # main.py:270
        loss.backward()
        names = [k for k, _ in net.named_parameters()]
        gproj, ghead = [], []
        for i, (k, p) in enumerate(net.named_parameters()):
            pr, hd = project(i, p.grad.numpy())
            gproj.append(pr)
            ghead.append(hd[:16] if hd.size >= 16 else np.pad(hd, (0, 16 - hd.size)))
        opt.step()
        pproj = [project(100 + i, p.detach().numpy() - sd[k])[0] for i, (k, p) in enumerate(net.named_parameters())]
        arrs["c%d_cfg" % ci] = np.array([int(online), nb, npair, nf, nt, seed, 1800 + ci, 1810 + ci, 1820 + ci])
        arrs["c%d_loss" % ci] = np.array(float(loss.detach()))
        arrs["c%d_pred" % ci] = pred.detach().numpy()
        arrs["c%d_gproj" % ci] = np.stack(gproj)
        arrs["c%d_ghead" % ci] = np.stack(ghead)
        arrs["c%d_dproj" % ci] = np.stack(pproj)
        arrs["c%d_names" % ci] = np.array(names)
        print("case", ci, "loss", float(loss.detach()))
    path = os.path.join(HERE, "g13_train.npz")
    np.savez(path, **arrs)
    print("g13_train %.1f KB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
