#!/usr/bin/env python3
"""G16: DP-IPD training TARGETS from the REAL reference (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_targets.py

Calls the reference's own ``DPIPD.forward(source_doa)`` (FN-SSL/Lightning/Module.py:464-498) and applies the ~25 lines of
``MyModel.data_preprocess`` that turn its output into gt_batch['ipd'] (FN-SSL/Lightning/main.py:227-262: real | imag of
bins 1..256, VAD mean over the segment's frames, threshold 0, mask, sum over sources) with the torch / numpy calls the
reference makes, in its order (main.py itself cannot be imported: pytorch_lightning is absent).  Only data is written.
"""
import os
import sys
import types
from copy import deepcopy

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/FN-SSL/Lightning"
sys.path.insert(0, REF)
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import Module as ref_module  # noqa: E402  (reference)

FRE_RANGE_USED = range(1, 257, 1)                                  # main.py:130


def ref_targets(gerdpipd, doa, vad, tar_use_vad):
    """main.py:227-262 (gt half of data_preprocess)."""
    source_doa = doa.cpu().numpy()
    _, ipd_batch, _ = gerdpipd(source_doa=source_doa)
    ipd_batch = np.concatenate((ipd_batch.real[:, :, FRE_RANGE_USED, :, :], ipd_batch.imag[:, :, FRE_RANGE_USED, :, :]),
                               axis=2).astype(np.float32)
    ipd_batch = torch.from_numpy(ipd_batch)
    vad_batch = vad.mean(axis=2).float()
    if tar_use_vad:
        nb, nt, nf, nmic, num_source = ipd_batch.shape
        th = 0
        vad_batch_copy = deepcopy(vad_batch)
        vad_batch_copy[vad_batch_copy <= th] = th
        vad_batch_copy[vad_batch_copy > 0] = 1
        vad_batch_expand = vad_batch_copy[:, :, np.newaxis, np.newaxis, :].expand(nb, nt, nf, nmic, num_source)
        ipd_batch = ipd_batch * vad_batch_expand
    ipd_batch = torch.sum(ipd_batch, dim=-1)
    return ipd_batch.numpy(), vad_batch.numpy()


def main():
    arrs = {}
    mics2 = np.array(((-0.04, 0.0, 0.0), (0.04, 0.0, 0.0)))        # main.py:121-123
    mics4 = np.array(((-0.04, 0.0, 0.0), (0.04, 0.0, 0.0), (0.0, 0.05, 0.01), (0.02, -0.03, 0.0)))
    cases = [("c0", mics2, "MM", 1, 3, 5, True), ("c1", mics4, "MM", 2, 2, 4, True), ("c2", mics4, "M", 2, 2, 3, False),
             ("c3", mics4, "MM", 3, 1, 2, True)]
    for name, mics, mode, ns, nb, nseg, use_vad in cases:
        rs = np.random.RandomState(1600 + len(name) + ns + nb)
        doa = np.stack((rs.uniform(0.2, np.pi - 0.2, (nb, nseg, ns)), rs.uniform(-np.pi, np.pi, (nb, nseg, ns))), axis=2)
        vad = (rs.uniform(size=(nb, nseg, 12, ns)) > 0.55).astype(np.float32)
        vad[0, 0, :, 0] = 0.0                                       # a silent source in one segment: its target is masked
        g = ref_module.DPIPD(ndoa_candidate=[5, 9], mic_location=mics, nf=257, fre_max=8000, ch_mode=mode, speed=340)
        ipd, vmean = ref_targets(g, torch.from_numpy(doa.astype(np.float32)), torch.from_numpy(vad), use_vad)
        _, raw, _ = g(source_doa=doa.astype(np.float32))
        arrs[name + "_mics"], arrs[name + "_doa"], arrs[name + "_vad"] = mics, doa.astype(np.float32), vad
        arrs[name + "_cfg"] = np.array([{"MM": 1, "M": 0}[mode], int(use_vad)])
        arrs[name + "_ipd"], arrs[name + "_vmean"] = ipd, vmean
        arrs[name + "_dpipd_sub"] = raw[:, :, ::32].astype(np.complex64)       # DPIPD.forward's own output, sub-sampled bins
    np.savez_compressed(os.path.join(HERE, "g16_dpipd_targets.npz"), **arrs)
    print("wrote g16_dpipd_targets.npz", {k: v.shape for k, v in arrs.items() if k.startswith("c1")})


if __name__ == "__main__":
    main()
