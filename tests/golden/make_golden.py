#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own modules
(/root/reference/FN-SSL/Lightning/{Model,Module,utils_}.py), feeds them seeded
inputs and weights from ``fnssl.weights`` and stores inputs' seeds/shapes and the
reference OUTPUTS as small .npz fixtures.  Only data is written here: no
reference source is copied.  The ~25 lines of ``main.py`` that stitch the front
end together (predict_step / data_preprocess, main.py:184-189,200-225) cannot be
imported (pytorch_lightning is absent) and are restated below by *calling the
reference's own classes in the reference's order*.

Inputs are regenerated in the tests from ``np.random.RandomState(seed)``.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/FN-SSL/Lightning"
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
sys.path.insert(0, REF)
sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))  # utils_.py:6 imports it

import numpy as np  # noqa: E402
import torch  # noqa: E402

import Model as ref_model  # noqa: E402  (reference)
import Module as ref_module  # noqa: E402  (reference)
import utils_ as ref_utils  # noqa: E402  (reference)
from fnssl import weights as W  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)


def rs_randn(seed, shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(np.float32)


def to_torch_sd(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def ref_data_preprocess(mic_sig, ch_mode, sample_length=298, eps=1e-6):
    """main.py:200-225 using the reference's STFT / AddChToBatch / forgetting_norm."""
    dostft = ref_module.STFT(win_len=512, win_shift_ratio=0.5, nfft=512)
    addbatch = ref_module.AddChToBatch(ch_mode=ch_mode)
    stft = dostft(signal=mic_sig)
    stft = stft.permute(0, 3, 1, 2)
    reb = addbatch(stft)
    mag = torch.abs(reb)
    mean_value = ref_utils.forgetting_norm(mag, sample_length)
    re = torch.real(reb) / (mean_value + eps)
    im = torch.imag(reb) / (mean_value + eps)
    x = torch.cat((re, im), dim=1)
    return x[:, :, range(1, 257), :], stft, reb, mean_value


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


@torch.no_grad()
def main():
    # ---- G1 STFT ---------------------------------------------------------
    sig = rs_randn(101, (2, 512 + 23 * 256, 4))
    stft = ref_module.STFT(512, 0.5, 512)(signal=torch.from_numpy(sig)).numpy()
    save("g1_stft", seed=101, shape=np.array(sig.shape), out=stft.astype(np.complex64))

    # ---- G2 AddChToBatch ---------------------------------------------------
    r = np.random.RandomState(102)
    d = (r.standard_normal((2, 4, 3, 2)) + 1j * r.standard_normal((2, 4, 3, 2))).astype(np.complex64)
    mm = ref_module.AddChToBatch("MM")(torch.from_numpy(d)).numpy()
    m = ref_module.AddChToBatch("M")(torch.from_numpy(d)).numpy()
    save("g2_pairs", inp=d, out_mm=mm, out_m=m)

    # ---- G3 forgetting_norm ------------------------------------------------
    mag = np.abs(rs_randn(103, (3, 2, 257, 24))) + 0.1
    o8 = ref_utils.forgetting_norm(torch.from_numpy(mag), 8).numpy()
    o298 = ref_utils.forgetting_norm(torch.from_numpy(mag), 298).numpy()
    mag_long = np.abs(rs_randn(104, (2, 2, 5, 310))) + 0.1
    olong = ref_utils.forgetting_norm(torch.from_numpy(mag_long), 298).numpy()
    save("g3_fnorm", seed=103, shape=np.array(mag.shape), out_sl8=o8, out_sl298=o298,
         seed_long=104, shape_long=np.array(mag_long.shape), out_long=olong)

    # ---- G4 features ---------------------------------------------------------
    sig3 = rs_randn(105, (1, 512 + 23 * 256, 3))
    x3, _, _, mu3 = ref_data_preprocess(torch.from_numpy(sig3), "MM")
    sig4 = rs_randn(106, (2, 512 + 23 * 256, 4), scale=0.05)
    x4, _, _, mu4 = ref_data_preprocess(torch.from_numpy(sig4), "MM")
    x4m, _, _, mu4m = ref_data_preprocess(torch.from_numpy(sig4), "M")
    save("g4_features", seed3=105, shape3=np.array(sig3.shape), x3=x3.numpy(), mu3=mu3.numpy(),
         seed4=106, shape4=np.array(sig4.shape), scale4=0.05,
         x4_sub=x4.numpy()[:, :, ::16, :], mu4=mu4.numpy(),
         x4m_sub=x4m.numpy()[:, :, ::16, :], mu4m=mu4m.numpy())

    # ---- G5 single LSTMs -------------------------------------------------------
    arrs = {}
    cases = [(4, 16, True), (36, 32, False), (4, 128, True), (256, 128, True),
             (260, 256, False), (256, 256, False), (260, 128, True)]
    for ci, (I, H, bi) in enumerate(cases):
        sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(I, H, bi)], seed=500 + ci)
        mod = torch.nn.LSTM(I, H, batch_first=True, bidirectional=bi)
        mod.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in sd.items()})
        x = rs_randn(600 + ci, (5, 7, I))
        y, _ = mod(torch.from_numpy(x))
        arrs["case%d_cfg" % ci] = np.array([I, H, int(bi), 500 + ci, 600 + ci, 5, 7])
        arrs["case%d_out" % ci] = y.numpy()
    save("g5_lstm", **arrs)

    # ---- G6 FNblock ------------------------------------------------------------
    arrs = {}
    for oi, online in enumerate([True, False]):
        sd1 = W.make_fnblock_state(700 + oi, 4, 32, online, True)
        b1 = ref_model.FNblock(4, hidden_size=32, is_online=online, is_first=True).eval()
        b1.load_state_dict(to_torch_sd(sd1))
        x = rs_randn(710 + oi, (2, 6, 8, 4))
        y1, fb1, nb1 = b1(torch.from_numpy(x))
        sd2 = W.make_fnblock_state(720 + oi, 32, 32, online, False)
        b2 = ref_model.FNblock(32, hidden_size=32, is_online=online, is_first=False).eval()
        b2.load_state_dict(to_torch_sd(sd2))
        y2, fb2, nb2 = b2(y1, fb_skip=fb1, nb_skip=nb1)
        t = "on" if online else "off"
        arrs.update({t + "_seeds": np.array([700 + oi, 710 + oi, 720 + oi]),
                     t + "_y1": y1.contiguous().numpy(), t + "_fb1": fb1.numpy(), t + "_nb1": nb1.numpy(),
                     t + "_y2": y2.contiguous().numpy(), t + "_fb2": fb2.numpy(), t + "_nb2": nb2.numpy()})
    save("g6_fnblock", **arrs)

    # ---- G7 / G8 / G11 FN_SSL -----------------------------------------------------
    arrs = {}
    for oi, online in enumerate([True, False]):
        sd = W.make_fnssl_state(800 + oi, is_online=online)
        net = ref_model.FN_SSL(is_online=online).eval()
        net.load_state_dict(to_torch_sd(sd))
        t = "on" if online else "off"
        xa = rs_randn(810 + oi, (2, 4, 16, 24))
        arrs[t + "_a"] = net(torch.from_numpy(xa)).numpy()
        xb = rs_randn(820 + oi, (1, 4, 256, 36))
        arrs[t + "_b"] = net(torch.from_numpy(xb)).numpy()
        xc = rs_randn(830 + oi, (3, 4, 16, 29))          # G11: nt % 12 != 0
        arrs[t + "_c"] = net(torch.from_numpy(xc)).numpy()
        arrs[t + "_seeds"] = np.array([800 + oi, 810 + oi, 820 + oi, 830 + oi])
    sd = W.make_fnssl_state(840, is_online=True, is_doa=True)
    net = ref_model.FN_SSL(is_online=True, is_doa=True).eval()
    net.load_state_dict(to_torch_sd(sd))
    xd = rs_randn(841, (1, 4, 256, 12))
    arrs["doa_out"] = net(torch.from_numpy(xd)).numpy()
    arrs["doa_seeds"] = np.array([840, 841])
    save("g7_fnssl", **arrs)

    # ---- G9 config 1 end to end -----------------------------------------------------
    sd = W.make_fnssl_state(900, is_online=True)
    net = ref_model.FN_SSL().eval()
    net.load_state_dict(to_torch_sd(sd))
    batch = rs_randn(901, (1, 2, 64000), scale=0.05)              # [nb, nch, ns]
    x, _, _, _ = ref_data_preprocess(torch.from_numpy(batch).permute(0, 2, 1), "MM")
    out = net(x)
    save("g9_config1", seeds=np.array([900, 901]), scale=0.05, shape=np.array(batch.shape),
         out=out.numpy(), x_sub=x.numpy()[:, :, ::32, ::8])




@torch.no_grad()
def golden_doa():
    """G12: IPD -> DOA back end (PredDOA.predgt2DOA / SourceDetectLocalize / DPIPD)."""
    arrs = {}
    # the reference's own 2-mic geometry, one and two sources, both source-number modes
    for ci, (ns, mode) in enumerate([(1, "kNum"), (2, "unkNum")]):
        pd = ref_module.PredDOA(method_mode="IDL", source_num_mode=mode, max_num_sources=ns, ch_mode="MM",
                                device="cpu")
        pred = np.tanh(rs_randn(1200 + ci, (3, 5, 512)))            # [nb*np (np = 1), nt, 2nf]
        out, _ = pd.predgt2DOA(pred_batch=torch.from_numpy(pred), gt_batch=None)
        arrs["c%d_cfg" % ci] = np.array([1200 + ci, 3, 5, ns, int(mode == "kNum")])
        arrs["c%d_doa" % ci] = out["doa"].numpy()
        arrs["c%d_vad" % ci] = out["vad_sources"].numpy()
        arrs["c%d_ss" % ci] = out["spatial_spectrum"].numpy()
    # template generator on a 4-mic geometry, 'MM' and 'M' (sub-sampled grid to stay small)
    mics = np.array(((-0.04, 0.0, 0.0), (0.04, 0.0, 0.0), (0.0, 0.05, 0.01), (0.02, -0.03, 0.0)))
    for mode in ("MM", "M"):
        g = ref_module.DPIPD(ndoa_candidate=[5, 9], mic_location=mics, nf=257, fre_max=8000, ch_mode=mode, speed=340)
        t, _, cand = g()
        arrs["tmpl_" + mode] = t[:, :, ::16, :].astype(np.complex64)
    arrs["tmpl_mics"] = mics
    # a 4-mic localisation through SourceDetectLocalize with a bank built by the reference DPIPD
    g = ref_module.DPIPD(ndoa_candidate=[37, 73], mic_location=mics, nf=257, fre_max=8000, ch_mode="MM", speed=340)
    t, _, _ = g()
    bank = np.concatenate((t.real[:, :, 1:257, :], t.imag[:, :, 1:257, :]), axis=2).astype(np.float32)
    bank = bank[18:19, 36:73]
    cand = [np.linspace(np.pi / 2, np.pi / 2, 1), np.linspace(0, np.pi, 37)]
    sdl = ref_module.SourceDetectLocalize(max_num_sources=2, source_num_mode="unkNum", meth_mode="IDL")
    pred4 = np.tanh(rs_randn(1210, (2, 4, 512, 6)))                # [nb, nt, 2nf, np]
    doa, vad, ss = sdl(pred_ipd=torch.from_numpy(pred4), dpipd_template=torch.from_numpy(bank), doa_candidate=cand)
    arrs["m4_seed"] = np.array([1210])
    arrs["m4_doa"], arrs["m4_vad"], arrs["m4_ss"] = doa.numpy(), vad.numpy(), ss.numpy()
    save("g12_doa", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "doa":
        golden_doa()
    else:
        main()
        golden_doa()
