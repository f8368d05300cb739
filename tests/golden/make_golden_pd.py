#!/usr/bin/env python3
"""G17: the 'PD' (peak detection) branch of SourceDetectLocalize from the REAL reference (build container only: needs
/root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pd.py

Calls /root/reference/FN-SSL/Lightning/Module.py::SourceDetectLocalize(meth_mode='PD') (Module.py:580-622) on seeded
[cos | sin] IPD vectors against template banks built by the reference's own DPIPD on a 4-mic geometry, and stores the
seeds / shapes and the reference OUTPUTS (doa, vad, spatial spectrum).  Only data is written; no reference source is copied.
The tests rebuild the inputs from the seeds and the banks with fnssl.doa.dpipd_templates (pinned by G12)."""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/FN-SSL/Lightning")
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import Module as ref_module  # noqa: E402  (reference)

MICS = np.array(((-0.04, 0.0, 0.0), (0.04, 0.0, 0.0), (0.0, 0.05, 0.01), (0.02, -0.03, 0.0)))


def rs_randn(seed, shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(np.float32)


@torch.no_grad()
def main():
    arrs = {"mics": MICS}
    cases = [  # (seed, nb, nt, nele, nazi, ch_mode, ns, source_num_mode, mix)
        (1700, 2, 4, 9, 19, "MM", 2, "unkNum", 0.0),
        (1701, 1, 5, 7, 13, "M", 2, "kNum", 0.0),
        # a true source direction mixed into the noise: the strongest peak is a real one
        (1703, 1, 4, 9, 19, "MM", 2, "unkNum", 1.5),
    ]
    # what the reference does for any other number of sources: the slice assignment at Module.py:615 raises
    raises = []
    for ns in (1, 3):
        g = ref_module.DPIPD(ndoa_candidate=[9, 19], mic_location=MICS, nf=257, fre_max=8000, ch_mode="MM", speed=340)
        t, _, cand = g()
        bank = np.concatenate((t.real[:, :, 1:257, :], t.imag[:, :, 1:257, :]), axis=2).astype(np.float32)
        sdl = ref_module.SourceDetectLocalize(max_num_sources=ns, source_num_mode="kNum", meth_mode="PD")
        try:
            sdl(pred_ipd=torch.from_numpy(np.tanh(rs_randn(1700, (2, 4, 512, 6)))), dpipd_template=torch.from_numpy(bank), doa_candidate=cand)
            raises.append(0)
        except RuntimeError:
            raises.append(1)
    arrs["raises_for_ns_1_3"] = np.array(raises)
    print("reference raises for ns = 1, 3:", raises)
    for ci, (seed, nb, nt, nele, nazi, mode, ns, snm, mix) in enumerate(cases):
        g = ref_module.DPIPD(ndoa_candidate=[nele, nazi], mic_location=MICS, nf=257, fre_max=8000, ch_mode=mode, speed=340)
        t, _, cand = g()
        bank = np.concatenate((t.real[:, :, 1:257, :], t.imag[:, :, 1:257, :]), axis=2).astype(np.float32)
        npair = bank.shape[-1]
        pred = np.tanh(rs_randn(seed, (nb, nt, 512, npair)))
        if mix:
            pred = (pred * 0.3 + mix * bank[nele // 2 + 1, 4][None, None]).astype(np.float32)
        sdl = ref_module.SourceDetectLocalize(max_num_sources=ns, source_num_mode=snm, meth_mode="PD")
        doa, vad, ss = sdl(pred_ipd=torch.from_numpy(pred), dpipd_template=torch.from_numpy(bank), doa_candidate=cand)
        arrs["c%d_cfg" % ci] = np.array([seed, nb, nt, nele, nazi, int(mode == "MM"), ns, int(snm == "kNum")])
        arrs["c%d_mix" % ci] = np.array([mix], dtype=np.float64)
        arrs["c%d_doa" % ci], arrs["c%d_vad" % ci], arrs["c%d_ss" % ci] = doa.numpy(), vad.numpy(), ss.numpy()
        arrs["c%d_cand_ele" % ci], arrs["c%d_cand_azi" % ci] = np.asarray(cand[0]), np.asarray(cand[1])
        print("case %d: doa %s vad %s" % (ci, doa.shape, vad.shape))
    arrs["ncases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(HERE, "g17_doa_pd.npz"), **arrs)
    print("wrote g17_doa_pd.npz")


if __name__ == "__main__":
    main()
