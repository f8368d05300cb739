"""GPU parity tests of the training path: reserve-saving LSTM forward, BPTT kernel,
gradient assembly, loss, optimizer — against PyTorch CPU autograd (oracle/train_ref.py, pinned to the real
reference by tests/golden/g13_train.npz).  fp32; gradients are compared relative to the tensor's largest
entry (sums over up to 1e5 terms in a different order)."""
import numpy as np
import pytest

from conftest import assert_close, load_golden, rs_randn

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; none visible (the HIP path has no CPU fallback)")
    from fnssl import _lib
    _lib.load()
    return torch.device("cuda:0")


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def rel_close(got, want, tol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want).max() / scale
    assert err <= tol, "%s: max err %.3g of the largest entry (tol %g)" % (what, err, tol)


@pytest.mark.parametrize("mode,H,bidir,c0,c2,c0g,nb,nt,nf", [
    ("narrow", 256, False, 256, 0, 256, 1, 7, 40),      # blocks 2/3 narrow-band, ragged group (40 = 2*16 + 8)
    ("narrow", 256, False, 256, 4, 256, 2, 5, 16),      # block 1 narrow-band: concat data channels, no grad for them
    ("full", 128, True, 256, 0, 256, 1, 18, 6),         # blocks 2/3 full-band, both directions
    ("full", 128, True, 0, 4, 0, 2, 9, 5),              # block 1 full-band: data only, no input gradient
    ("narrow", 128, True, 256, 0, 256, 1, 6, 20),       # offline narrow-band
])
def test_lstm_layer_gradients_match_autograd(dev, mode, H, bidir, c0, c2, c0g, nb, nt, nf):
    from fnssl import ops
    from fnssl import weights as W
    ndir = 2 if bidir else 1
    I = c0 + c2
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(I, H, bidir)], seed=900 + H + I)
    x = rs_randn(901, (nb, nt, nf, I), 0.7)
    gout = rs_randn(902, (nb, nt, nf, ndir * H), 1.0)
    # ---- oracle: nn.LSTM + autograd on the CPU
    lstm = torch.nn.LSTM(I, H, batch_first=True, bidirectional=bidir)
    lstm.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in sd.items()})
    xt = torch.from_numpy(x).requires_grad_(True)
    seq = xt.reshape(nb * nt, nf, I) if mode == "full" else xt.permute(0, 2, 1, 3).reshape(nb * nf, nt, I)
    y, _ = lstm(seq)
    yl = y.reshape(nb, nt, nf, -1) if mode == "full" else y.reshape(nb, nf, nt, -1).permute(0, 2, 1, 3)
    (yl * torch.from_numpy(gout)).sum().backward()
    # ---- HIP: forward with reserve, BPTT, weight gradients as GEMMs on the kernel's dA
    sfx = [""] + (["_reverse"] if bidir else [])
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s],
                            sd["L.bias_hh_l0" + s], c0, c2, dev) for s in sfx]
    packed_b = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev)
                for s in sfx]

    def natural(shape_c):   # a logical [nb, nt, nf, C] tensor stored in the layer's natural layout
        if mode == "full":
            return torch.zeros((nb, nt, nf, shape_c), device=dev)
        return torch.zeros((nb, nf, nt, shape_c), device=dev).permute(0, 2, 1, 3)

    xd = to_dev(x, dev)
    x0 = xd[..., :c0].contiguous() if c0 else None
    x2 = xd[..., c0:].contiguous() if c2 else None
    out = natural(ndir * H)
    nseq, nsteps = (nb * nt, nf) if mode == "full" else (nb * nf, nt)
    reserve = torch.empty(ops.lstm_reserve_floats(nseq, H, ndir, nsteps), device=dev)
    ops.lstm_layer(mode, x0, None, x2, packed, H, out, reserve=reserve)
    assert_close(out.cpu().numpy(), yl.detach().numpy(), 1e-4, 1e-5, "training forward")
    plain = natural(ndir * H)
    ops.lstm_layer(mode, x0, None, x2, packed, H, plain)
    assert torch.equal(out, plain), "the reserve-saving forward computes the same h"
    da = natural(ndir * 4 * H)
    dx = natural(ndir * c0g) if c0g else None
    ops.lstm_backward(mode, reserve, to_dev(gout, dev), da, dx, packed_b, H, c0g)
    if c0g:
        dxs = dx.reshape(nb, nt, nf, ndir, c0g).sum(dim=3)
        rel_close(dxs.cpu().numpy(), xt.grad.numpy()[..., :c0g], 2e-4, "dx")
    # rows in sequence-major order: [seq, step, ...]
    def rows(t):
        return t.reshape(nseq, nsteps, -1) if mode == "full" else t.permute(0, 2, 1, 3).reshape(nseq, nsteps, -1)
    dar, xr, hr = rows(da), rows(xd), rows(out)
    for di, s in enumerate(sfx):
        a = dar[..., di * 4 * H:(di + 1) * 4 * H].reshape(-1, 4 * H)
        hprev = torch.zeros((nseq, nsteps, H), device=dev)
        hd = hr[..., di * H:(di + 1) * H]
        if di == 0:
            hprev[:, 1:] = hd[:, :-1]
        else:
            hprev[:, :-1] = hd[:, 1:]
        dwih = a.t() @ xr.reshape(-1, I)
        dwhh = a.t() @ hprev.reshape(-1, H)
        db = a.sum(dim=0)
        g = {n: p.grad.numpy() for n, p in lstm.named_parameters()}
        rel_close(dwih.cpu().numpy(), g["weight_ih_l0" + s], 2e-4, "dW_ih" + s)
        rel_close(dwhh.cpu().numpy(), g["weight_hh_l0" + s], 2e-4, "dW_hh" + s)
        rel_close(db.cpu().numpy(), g["bias_ih_l0" + s], 2e-4, "db" + s)


def _engine(dev, online, wseed, **kw):
    import Model
    from fnssl import train
    from fnssl import weights as W
    sd = W.make_fnssl_state(wseed, 4, 256, online)
    net = Model.FN_SSL(is_online=online)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.to(dev)
    return sd, net, train.TrainEngine(net, **kw)


def test_dropout_scale_matches_oracle_hash(dev):
    from fnssl import train
    from oracle import train_ref as T
    for seed, layer in ((0, 0), (123456789, 5)):
        s32 = train.layer_seed(seed, layer)
        assert s32 == T.layer_seed(seed, layer)
        got = train.dropout_scale((3, 5, 7, 256), s32, dev, b0=2).cpu().numpy()
        assert np.array_equal(got, T.dropout_scale(s32, (3, 5, 7, 256), b0=2))


def test_combine_is_dropout_plus_residual_in_any_layout(dev):
    from fnssl import train
    from oracle import train_ref as T
    nb, nt, nf, c = 2, 5, 6, 256
    a = to_dev(rs_randn(1, (nb, nt, nf, c)), dev)
    b_store = to_dev(rs_randn(2, (nb, nf, nt, c)), dev)          # narrow-band storage
    cc = to_dev(rs_randn(3, (nb, nt, nf, c)), dev)
    s32 = train.layer_seed(9, 2)
    out = torch.empty((nb, nf, nt, c), device=dev).permute(0, 2, 1, 3)
    train.combine(out, masked=(a, b_store.permute(0, 2, 1, 3)), plain=(cc,), seed32=s32, b0=4)
    m = T.dropout_scale(s32, (nb, nt, nf, c), b0=4)
    want = (a.cpu().numpy() + b_store.permute(0, 2, 1, 3).cpu().numpy()) * m + cc.cpu().numpy()
    assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))
    plain = torch.empty((nb, nt, nf, c), device=dev)
    train.combine(plain, plain=(a, cc))
    assert torch.equal(plain, a + cc)


@pytest.mark.parametrize("case", [0, 1])
def test_training_step_matches_oracle_and_reference_golden(dev, case):
    """Loss, every gradient and the Adam update of one step against the autograd oracle, and against the
    projections of the REAL reference's gradients (g13)."""
    from oracle import train_ref as T
    g = load_golden("g13_train")
    online, nb, npair, nf, nt, seed, wseed, xseed, gseed = [int(v) for v in g["c%d_cfg" % case]]
    sd, net, eng = _engine(dev, bool(online), wseed)
    x = rs_randn(xseed, (nb * npair, 4, nf, nt))
    gt = rs_randn(gseed, (nb, nt // 12, 2 * nf, npair), 0.5)
    # make the engine draw the golden's masks
    eng.seed, eng.step_count = 0, 0
    import fnssl.train as tr
    orig = tr.layer_seed
    try:
        tr.layer_seed = lambda base, l: orig(seed, l)
        loss = eng.step(to_dev(x, dev), to_dev(gt, dev))
    finally:
        tr.layer_seed = orig
    want_loss, grads, new_sd, _, _ = T.train_step(sd, x, gt, seed, 256, bool(online))
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    assert abs(loss - float(g["c%d_loss" % case])) <= 1e-5 * abs(want_loss)
    got = eng.gradients()
    names = [str(s) for s in g["c%d_names" % case]]
    assert names == list(got.keys())
    for i, k in enumerate(names):
        rel_close(got[k].cpu().numpy(), grads[k], 5e-4, "grad " + k)
        a = got[k].cpu().numpy().astype(np.float64)
        r = rs_randn(5000 + i, a.shape).astype(np.float64)
        want = g["c%d_gproj" % case][i]
        assert abs(np.sqrt((a * a).sum()) - want[0]) <= 5e-4 * want[0] + 1e-12, ("norm vs reference", k)
    for k in names:
        d_got = net.state_dict()[k].cpu().numpy() - sd[k]
        d_want = new_sd[k] - sd[k]
        # Adam's first step moves every weight by ~lr * sign(grad): compare where the gradient is not tiny
        big = np.abs(grads[k]) > 1e-3 * np.abs(grads[k]).max()
        assert np.abs(d_got - d_want)[big].max() <= 2e-5, ("adam update", k)


def test_training_steps_chunked_equals_whole_and_adam_state_advances(dev):
    from oracle import train_ref as T
    nb, npair, nf, nt = 2, 2, 4, 12
    x = rs_randn(31, (nb * npair, 4, nf, nt))
    gt = rs_randn(32, (nb, 1, 2 * nf, npair), 0.5)
    sd, net_a, eng_a = _engine(dev, True, 77, seed=5)
    _, net_b, eng_b = _engine(dev, True, 77, seed=5, chunk_pairs=2)
    la = [eng_a.step(to_dev(x, dev), to_dev(gt, dev)) for _ in range(2)]
    lb = [eng_b.step(to_dev(x, dev), to_dev(gt, dev)) for _ in range(2)]
    assert abs(la[0] - lb[0]) <= 1e-6 * abs(la[0]) and abs(la[1] - lb[1]) <= 1e-5 * abs(la[1])
    for k, v in net_a.state_dict().items():
        assert (v - net_b.state_dict()[k]).abs().max() <= 1e-5, k
    # two oracle steps with the engine's seeds
    s1 = (5 * 1000003 + 1 * 8191) & 0xFFFFFFFF
    s2 = (5 * 1000003 + 2 * 8191) & 0xFFFFFFFF
    l1, _, sd1, st1, _ = T.train_step(sd, x, gt, s1, 256, True)
    l2, _, sd2, _, _ = T.train_step(sd1, x, gt, s2, 256, True, adam_state=st1, step=2)
    assert abs(la[0] - l1) <= 1e-5 * abs(l1) and abs(la[1] - l2) <= 1e-4 * abs(l2)
    assert la[1] < la[0], "the loss goes down on a repeated batch"


def test_mymodel_training_step_dropin(dev):
    """Lightning-style surface: training_step((mic_sig, {'ipd': gt})) -> {'loss'}; the inference entry sees
    the updated weights afterwards; cal_loss equals the oracle's MSE."""
    import predict_step as ps
    from fnssl import weights as W
    from oracle import train_ref as T
    m = ps.MyModel(device="cuda")
    m.arch.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in W.make_fnssl_state(5).items()})
    m.to(dev)
    sig = to_dev(rs_randn(41, (2, 512 + 23 * 256, 2), 0.1), dev)          # [nb, ns, nch] -> 24 frames
    gt = to_dev(np.full((2, 2, 512, 1), 0.5, dtype=np.float32) + rs_randn(42, (2, 2, 512, 1), 0.05), dev)   # learnable
    pred0 = m.predict_step(sig.permute(0, 2, 1))
    l0 = m.cal_loss(pred0, {"ipd": gt})
    want = float(T.cal_loss(pred0.cpu(), gt.cpu()))
    assert abs(float(l0) - want) <= 1e-6 * want
    out = m.training_step((sig, {"ipd": gt}), 0)
    assert out["loss"].shape == () and torch.isfinite(out["loss"])
    pred1 = m.predict_step(sig.permute(0, 2, 1))
    assert not torch.equal(pred0, pred1), "inference must pick up the trained weights"
    for _ in range(3):
        m.training_step((sig, {"ipd": gt}), 0)
    assert float(m.cal_loss(m.predict_step(sig.permute(0, 2, 1)), {"ipd": gt})) < float(l0)
    opt = m.configure_optimizers()          # the arithmetic-free shim Lightning counts steps on (predict_step.EngineOptimizer)
    assert isinstance(opt, torch.optim.Optimizer) and opt.state_dict()["engine"]["step_count"] == 4


@pytest.mark.parametrize("mode,H,bidir,c0,c2,nb,nt,nf", [
    ("narrow", 256, False, 256, 0, 2, 3, 4096),      # 512 groups -> 4 waves per group, 8-wave workgroups
    ("narrow", 256, False, 256, 4, 2, 3, 4096),
    ("narrow", 256, False, 256, 0, 2, 3, 8192),      # 1024 groups -> 2 waves per group, 4-wave workgroups
    ("full", 128, True, 256, 0, 1, 9600, 3),         # config-4 full-band geometry: 600 groups x 2 directions
    ("full", 128, True, 4, 0, 1, 9600, 3),
])
def test_split_static_training_forward_equals_generic(dev, mode, H, bidir, c0, c2, nb, nt, nf):
    """The shape-specialised split kernels of the training forward write the same h AND the same reserve as the
    generic kernels, bit for bit (env FNSSL_TRAIN_NO_STATIC switches them off)."""
    import os
    from fnssl import ops
    from fnssl import weights as W
    ndir = 2 if bidir else 1
    I = c0 + c2
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(I, H, bidir)], seed=970 + c0 + c2)
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s],
                            sd["L.bias_hh_l0" + s], c0, c2, dev) for s in ([""] + (["_reverse"] if bidir else []))]
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn((nb, nt, nf, I), generator=g, device=dev) * 0.7
    x0, x2 = x[..., :c0].contiguous(), (x[..., c0:].contiguous() if c2 else None)
    nseq, nsteps = (nb * nt, nf) if mode == "full" else (nb * nf, nt)
    res = []
    for no_static in ("0", "1"):
        os.environ["FNSSL_TRAIN_NO_STATIC"] = no_static
        os.environ["FNSSL_TRAIN_NO_F32_CLUSTER"] = "1"       # (this test is about the split kernels; the cluster kernel has its own)
        ops._lib.refresh_tuning()
        try:
            out = torch.full((nb, nt, nf, ndir * H), float("nan"), device=dev)
            reserve = torch.full((ops.lstm_reserve_floats(nseq, H, ndir, nsteps),), float("nan"), device=dev)
            ops.lstm_layer(mode, x0, None, x2, packed, H, out, reserve=reserve)
            res.append((out, reserve))
        finally:
            os.environ.pop("FNSSL_TRAIN_NO_STATIC", None)
            os.environ.pop("FNSSL_TRAIN_NO_F32_CLUSTER", None)
            ops._lib.refresh_tuning()
    assert not torch.isnan(res[0][0]).any()
    assert torch.equal(res[0][0], res[1][0]), "h"
    assert torch.equal(torch.nan_to_num(res[0][1], nan=-7.0), torch.nan_to_num(res[1][1], nan=-7.0)), "reserve"


@pytest.mark.parametrize("c0,nb,nt,nf", [(256, 32, 300, 7), (4, 32, 300, 7), (256, 33, 301, 5)])
def test_cluster_training_forward_equals_split_kernels(dev, monkeypatch, c0, nb, nt, nf):
    """Round 4: the full-band layers of the training forward at config 4's shard (600 groups x 2 directions on 32 clusters:
    37.5 groups per cluster, the ones beyond two per wave rotating over the waves) run the cluster-resident kernel with the
    reserve stores (lstm_f32c.h, kSave) — h AND the reserve bit for bit equal to the 2-waves-per-group split kernels
    (FNSSL_TRAIN_NO_F32_CLUSTER=1), twice; the third case has a ragged last group and groups that cross utterances."""
    from fnssl import ops
    from fnssl import weights as W
    H = 128
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0, H, True)], seed=990 + c0)
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0, 0, dev)
              for s in ("", "_reverse")]
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.randn((nb, nt, nf, c0), generator=g, device=dev) * 0.7

    def run(plan=False):
        out = torch.full((nb, nt, nf, 2 * H), float("nan"), device=dev)
        reserve = torch.full((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), float("nan"), device=dev)
        r = ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve, plan_only=plan)
        return r if plan else (out, reserve)

    monkeypatch.delenv("FNSSL_TRAIN_NO_F32_CLUSTER", raising=False)
    assert run(plan=True)[0] == "f32_cluster", run(plan=True)
    a, ra = run()
    a2, ra2 = run()
    monkeypatch.setenv("FNSSL_TRAIN_NO_F32_CLUSTER", "1")
    assert run(plan=True)[0] == "train"
    b, rb = run()
    assert not torch.isnan(a).any()
    assert torch.equal(a, b) and torch.equal(a, a2), "h"
    nn = lambda t: torch.nan_to_num(t, nan=-7.0)  # noqa: E731  (rows of a ragged last group stay unwritten in both)
    assert torch.equal(nn(ra), nn(rb)) and torch.equal(nn(ra), nn(ra2)), "reserve"


@pytest.mark.parametrize("mode,H,c0,c2,nb,nt,nf", [
    ("narrow", 256, 256, 0, 16, 9, 256),      # 16 two-mic utterances per GPU: 256 groups = 16 per cluster of 16 (blocks 2-3)
    ("narrow", 256, 256, 4, 16, 7, 256),      # block 1: the concatenated data channels
    ("narrow", 256, 256, 4, 5, 11, 250),      # ragged: 1250 sequences, groups cross pairs
    ("full", 128, 256, 0, 16, 300, 5),        # a 16-utterance shard's full-band layer (600 groups: below the round-4 threshold)
    ("full", 128, 4, 0, 3, 40, 6),            # a tiny shard: 8 groups per direction
])
def test_cluster_training_forward_of_smaller_shards_equals_split_kernels(dev, monkeypatch, mode, H, c0, c2, nb, nt, nf):
    """Round 5: the reserve-saving forward of shards SMALLER than config 4's runs on the cluster-resident kernel too — the
    H = 256 narrow-band layers (clusters of 16, kSave form) below two groups per CU, the H = 128 full-band layers at any size.
    The several-waves-per-group kernels they replace took as long for 16 utterances as for 32 (85.9 / 38.3 ms against 59.9 /
    37.1).  h AND the reserve bit for bit equal to those kernels (TRAIN_NO_F32_CLUSTER), twice."""
    from fnssl import ops
    from fnssl import weights as W
    bidir = mode == "full"
    ndir = 2 if bidir else 1
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, bidir)], seed=995 + c0 + c2)
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0, c2, dev)
              for s in (("", "_reverse") if bidir else ("",))]
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    x0 = torch.randn((nb, nt, nf, c0), generator=g, device=dev) * 0.7
    x2 = torch.randn((nb, nt, nf, c2), generator=g, device=dev) * 0.7 if c2 else None
    nseq, nsteps = (nb * nt, nf) if mode == "full" else (nb * nf, nt)

    def run(plan=False):
        if mode == "full":
            out = torch.full((nb, nt, nf, ndir * H), float("nan"), device=dev)
        else:
            out = torch.full((nb, nf, nt, ndir * H), float("nan"), device=dev).permute(0, 2, 1, 3)
        reserve = torch.full((ops.lstm_reserve_floats(nseq, H, ndir, nsteps),), float("nan"), device=dev)
        r = ops.lstm_layer(mode, x0, None, x2, packed, H, out, reserve=reserve, plan_only=plan)
        return r if plan else (out, reserve)

    for k in ("FNSSL_TRAIN_NO_F32_CLUSTER", "FNSSL_NO_F32_SMALL", "FNSSL_NO_F32_CLUSTER"):
        monkeypatch.delenv(k, raising=False)
    ops.cluster_fallbacks(dev, reset=True)
    assert run(plan=True)[0] == "f32_cluster", run(plan=True)
    a, ra = run()
    a2, ra2 = run()
    assert ops.cluster_fallbacks(dev) == 0
    monkeypatch.setenv("FNSSL_TRAIN_NO_F32_CLUSTER", "1")
    assert run(plan=True)[0] == "train"
    b, rb = run()
    assert not torch.isnan(a).any()
    assert torch.equal(a, b) and torch.equal(a, a2), "h"
    nn = lambda t: torch.nan_to_num(t, nan=-7.0)  # noqa: E731  (rows of a ragged last group stay unwritten in both)
    assert torch.equal(nn(ra), nn(rb)) and torch.equal(nn(ra), nn(ra2)), "reserve"


@pytest.mark.parametrize("c0g,nb,nt,nf", [(256, 32, 300, 7), (256, 33, 301, 5), (0, 32, 300, 7), (256, 3, 40, 6), (0, 2, 300, 5)])
def test_cluster_bptt_equals_split_kernels(dev, monkeypatch, c0g, nb, nt, nf):
    """Round 4: back-propagation through time of the H = 128 full-band layers at config 4's shard on the cluster-resident
    kernel (lstm_bwdc.h: the 6 output slices of [W_ih | W_hh]^T over clusters of 6 CUs, 30 groups per cluster on 12 waves,
    the 6 beyond two per wave rotating) — dA AND dx bit for bit equal to the 2-waves-per-group split kernels
    (FNSSL_BWD_NO_CLUSTER=1), twice, status word 0; the second case has a ragged last group and groups that cross
    utterances; the third is block 1's layer (no input gradient: 2 output slices, clusters of 2 CUs, 4 hidden slices of
    gate gradients per member)."""
    from fnssl import ops
    from fnssl import weights as W
    H = 128
    c_in = c0g if c0g else 16
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c_in, H, True)], seed=770 + nb)
    sfx = ("", "_reverse")
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c_in, 0, dev)
              for s in sfx]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    x = torch.randn((nb, nt, nf, c_in), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 2 * H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
    ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)

    def run(plan=False):
        da = torch.full((nb, nt, nf, 2 * 4 * H), float("nan"), device=dev)
        dx = torch.full((nb, nt, nf, 2 * c0g), float("nan"), device=dev) if c0g else None
        if plan:
            return ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, plan_only=True)
        _, _, word = ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, status=True)
        return da, (dx if c0g else torch.zeros(1, device=dev)), word

    monkeypatch.delenv("FNSSL_BWD_NO_CLUSTER", raising=False)
    assert run(plan=True) == "bwd_cluster"
    a, xa, wa = run()
    a2, xa2, _ = run()
    monkeypatch.setenv("FNSSL_BWD_NO_CLUSTER", "1")
    assert run(plan=True) == "bwd"
    b, xb, wb = run()
    assert wa == 0 and wb == 0
    assert not torch.isnan(a).any() and not torch.isnan(xa).any()
    assert torch.equal(a, b) and torch.equal(a, a2), "dA"
    assert torch.equal(xa, xb) and torch.equal(xa, xa2), "dx"


def test_cluster_bptt_is_not_taken_for_layouts_its_addressing_cannot_express(dev):
    """The cluster BPTT kernel's per-lane offsets assume that a group's first sequence is its lowest address (dense
    [b, t] rows: so >= (q_inner - 1) * si).  fnssl_lstm_backward accepts any non-negative strides that are multiples of
    4, so a TIME-MAJOR upstream gradient / dA buffer (so < si) at a cluster-eligible size must go to the split kernels —
    whose 64-bit address search takes any layout — and give the dense layout's numbers bit for bit (round-4 advisor
    finding: bwdc_handles() checked none of this; the wrapped 32-bit offset would have gone out of bounds)."""
    from fnssl import ops
    from fnssl import weights as W
    H, c0g, nb, nt, nf = 128, 256, 32, 300, 5
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0g, H, True)], seed=811)
    sfx = ("", "_reverse")
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0g, 0, dev)
              for s in sfx]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    x = torch.randn((nb, nt, nf, c0g), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 2 * H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
    ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)
    # dense
    da = torch.full((nb, nt, nf, 8 * H), float("nan"), device=dev)
    dx = torch.full((nb, nt, nf, 2 * c0g), float("nan"), device=dev)
    assert ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, plan_only=True) == "bwd_cluster"
    ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g)
    # time-major storage [t, b, f, C] seen through the same logical [b, t, f, C] view
    tm = lambda t: t.permute(1, 0, 2, 3).contiguous().permute(1, 0, 2, 3)   # noqa: E731
    for which in ("dh", "da", "dx"):
        dh2 = tm(dh) if which == "dh" else dh
        da2 = torch.full((nt, nb, nf, 8 * H), float("nan"), device=dev).permute(1, 0, 2, 3) if which == "da" else torch.full_like(da, float("nan"))
        dx2 = torch.full((nt, nb, nf, 2 * c0g), float("nan"), device=dev).permute(1, 0, 2, 3) if which == "dx" else torch.full_like(dx, float("nan"))
        assert ops.lstm_backward("full", reserve, dh2, da2, dx2, bw, H, c0g, plan_only=True) == "bwd", which
        ops.lstm_backward("full", reserve, dh2, da2, dx2, bw, H, c0g)
        assert torch.equal(da2, da) and torch.equal(dx2, dx), which


def test_cluster_bptt_stays_resident_beside_a_tenant_that_holds_16_compute_units(dev):
    """Multi-GPU readiness without a second GPU (round-4 review): under DDP-style overlap RCCL's persistent all-reduce kernels
    hold some compute units while the BPTT of the lower layers runs, and the cluster-resident BPTT kernel wants EVERY member
    workgroup resident at once.  Stand-in for RCCL: fnssl_occupy_cus — 16 workgroups that each claim a whole CU's LDS and
    idle on a side stream while block 1's layer (clusters of 2: all 256 CUs by default) runs.
      * default geometry beside the tenant: the members that find no free CU move in when the first clusters finish —
        correct, no fallback, but the layer takes ~1.75 x as long (asserted: >= 1.3 x, i.e. the tenant does bite);
      * fnssl_tuning RESERVED_CUS = 16 (what TrainEngine puts on the backward's calls when world > 1): the launch is sized
        for the remaining CUs — same dA bit for bit, no fallback, within 10 % of the free device."""
    import time
    from fnssl import _lib, ops
    from fnssl import weights as W
    H, nb, nt, nf, c_in = 128, 28, 300, 64, 16                      # 525 groups per direction: 9 per cluster of 2
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c_in, H, True)], seed=1)
    sfx = ("", "_reverse")
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c_in, 0, dev)
              for s in sfx]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], 0)).to(dev) for s in sfx]
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    x = torch.randn((nb, nt, nf, c_in), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 2 * H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
    ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)
    side = torch.cuda.Stream(device=dev)
    stop = torch.zeros(1 + 16, dtype=torch.int32).pin_memory()

    def run(tenant, **knobs):
        da = torch.full((nb, nt, nf, 8 * H), float("nan"), device=dev)
        with _lib.tuning(**knobs):
            assert ops.lstm_backward("full", reserve, dh, da, None, bw, H, 0, plan_only=True) == "bwd_cluster"
            torch.cuda.synchronize()
            resident = 0
            if tenant:
                stop.zero_()
                ops.occupy_cus(16, stop, max_ms=2000, stream=side)
                t_end = time.time() + 1.0
                while int(stop[1:].sum()) < 16 and time.time() < t_end:
                    time.sleep(0.001)
                resident = int(stop[1:].sum())
            ops.cluster_fallbacks(dev, reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _, _, word = ops.lstm_backward("full", reserve, dh, da, None, bw, H, 0, status=True)
            e1.record()
            e1.synchronize()
            stop[0] = 1
            side.synchronize()
        assert word == 0 and ops.cluster_fallbacks(dev) == 0
        assert not tenant or resident == 16, "the stand-in tenant never became resident (%d of 16)" % resident
        return e0.elapsed_time(e1), da

    run(False)                                                     # warm-up
    best = lambda legs: min(legs, key=lambda r: r[0])   # noqa: E731   (best of three: launch jitter is not the subject)
    t_free, da_free = best([run(False) for _ in range(3)])
    t_busy, da_busy = best([run(True) for _ in range(3)])
    t_res_free, da_r = best([run(False, reserved_cus=16) for _ in range(3)])
    t_res_busy, da_rb = best([run(True, reserved_cus=16) for _ in range(3)])
    assert torch.equal(da_busy, da_free) and torch.equal(da_r, da_free) and torch.equal(da_rb, da_free)
    # measured: 4.4 against 2.5 ms (default geometry beside the tenant), 2.47 against 2.47 ms (sized for 240 CUs)
    assert t_busy >= 1.2 * t_free, "the tenant did not take CUs the default geometry needs (%.2f vs %.2f ms): nothing shown" % (t_busy, t_free)
    assert t_res_busy <= 1.15 * t_res_free + 0.2, (t_res_busy, t_res_free)
    assert t_res_free <= 1.15 * t_free + 0.2, "sizing for 240 CUs must not cost more than the 16 CUs it gives up (%.2f vs %.2f ms)" % (t_res_free, t_free)


@pytest.mark.parametrize("nb,nt,nf", [(32, 7, 256), (33, 5, 250), (31, 6, 256)])
def test_two_groups_per_wave_bptt_equals_one_group_per_wave_set(dev, monkeypatch, nb, nt, nf):
    """Round 4: the narrow-band layers' BPTT (H = 256, one direction, ~512 groups at config 4's shard) with both groups of a CU
    against ONE stream of weight records (lstm_bwd2.h) — dA and dx bit for bit equal to the 4-waves-per-group kernels that
    stream the matrix once per group (FNSSL_NO_BWD2=1), twice; ragged last group, groups that cross utterances, and an odd
    number of groups (the last workgroup has one)."""
    from fnssl import ops
    from fnssl import weights as W
    H, c0g, c2 = 256, 256, 4
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0g + c2, H, False)], seed=880 + nb)
    packed = [ops.pack_lstm(sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], sd["L.bias_ih_l0"], sd["L.bias_hh_l0"], c0g, c2, dev)]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], c0g)).to(dev)]
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x0 = torch.randn((nb, nt, nf, c0g), generator=g, device=dev) * 0.7
    x2 = torch.randn((nb, nt, nf, c2), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nf, H, 1, nt),), device=dev)
    ops.lstm_layer("narrow", x0, None, x2, packed, H, out, reserve=reserve)

    def run():
        da = torch.full((nb, nt, nf, 4 * H), float("nan"), device=dev)
        dx = torch.full((nb, nt, nf, c0g), float("nan"), device=dev)
        ops.lstm_backward("narrow", reserve, dh, da, dx, bw, H, c0g)
        return da, dx

    monkeypatch.delenv("FNSSL_NO_BWD2", raising=False)
    a, xa = run()
    a2, xa2 = run()
    monkeypatch.setenv("FNSSL_NO_BWD2", "1")
    b, xb = run()
    assert not torch.isnan(a).any() and not torch.isnan(xa).any()
    assert torch.equal(a, b) and torch.equal(a, a2), "dA"
    assert torch.equal(xa, xb) and torch.equal(xa, xa2), "dx"


@pytest.mark.parametrize("c2,nb,nt,nf", [(0, 32, 7, 256), (4, 32, 6, 256), (4, 33, 5, 250), (0, 31, 6, 256)])
def test_two_groups_per_wave_training_forward_equals_four_waves_per_group(dev, monkeypatch, c2, nb, nt, nf):
    """Round 4: the narrow-band layers' reserve-saving forward (H = 256, one direction, ~512 groups at config 4's shard) with
    both groups of a CU against ONE stream of weight records (lstm_fwd2.h; x_t and h_{t-1} through LDS, cell state in
    registers) — h AND the reserve bit for bit equal to the 4-waves-per-group kernels (FNSSL_NO_FWD2=1), twice; with and
    without block 1's 4-channel second input, ragged last group, an odd number of groups."""
    from fnssl import ops
    from fnssl import weights as W
    H, c0 = 256, 256
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, False)], seed=660 + nb + c2)
    packed = [ops.pack_lstm(sd["L.weight_ih_l0"], sd["L.weight_hh_l0"], sd["L.bias_ih_l0"], sd["L.bias_hh_l0"], c0, c2, dev)]
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    x0 = torch.randn((nb, nt, nf, c0), generator=g, device=dev) * 0.7
    x2 = torch.randn((nb, nt, nf, c2), generator=g, device=dev) * 0.7 if c2 else None

    def run():
        out = torch.full((nb, nt, nf, H), float("nan"), device=dev)
        reserve = torch.full((ops.lstm_reserve_floats(nb * nf, H, 1, nt),), float("nan"), device=dev)
        ops.lstm_layer("narrow", x0, None, x2, packed, H, out, reserve=reserve)
        return out, reserve

    # (end of round 5: the cluster-resident kernel with the streamed row takes this launch by default — lstm_f32c.h — and
    #  lstm_fwd2_kernel is its guarded fallback: all three families must agree)
    for k in ("FNSSL_NO_FWD2", "FNSSL_TRAIN_NO_F32_CLUSTER"):
        monkeypatch.delenv(k, raising=False)
    c, rc = run()
    monkeypatch.setenv("FNSSL_TRAIN_NO_F32_CLUSTER", "1")
    a, ra = run()
    a2, ra2 = run()
    monkeypatch.setenv("FNSSL_NO_FWD2", "1")
    b, rb = run()
    assert not torch.isnan(a).any()
    assert torch.equal(a, b) and torch.equal(a, a2) and torch.equal(a, c), "h"
    nn = lambda t: torch.nan_to_num(t, nan=-7.0)  # noqa: E731  (rows of a ragged last group stay unwritten in all)
    assert torch.equal(nn(ra), nn(rb)) and torch.equal(nn(ra), nn(ra2)) and torch.equal(nn(ra), nn(rc)), "reserve"


def test_cluster_bptt_gives_up_cleanly_and_the_same_call_recomputes_the_layer(dev, monkeypatch):
    """A member of cluster 0 that never shows up (FNSSL_CLUSTER_TEST_STALL): the waiting waves give up after the bounded
    number of spins, record a status word, every workgroup drains — no trap, the device stays usable — and the guarded
    fallback kernels of the same fnssl_lstm_backward call recompute dA and dx, bit-identical to the split kernels."""
    from fnssl import ops
    from fnssl import weights as W
    H, c0g, nb, nt, nf = 128, 256, 32, 300, 4
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0g, H, True)], seed=31)
    sfx = ("", "_reverse")
    packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c0g, 0, dev)
              for s in sfx]
    bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    x = torch.randn((nb, nt, nf, c0g), generator=g, device=dev) * 0.7
    dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 2 * H), device=dev)
    reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
    ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)

    def run():
        da = torch.full((nb, nt, nf, 2 * 4 * H), float("nan"), device=dev)
        dx = torch.full((nb, nt, nf, 2 * c0g), float("nan"), device=dev)
        return ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, status=True)

    monkeypatch.setenv("FNSSL_BWD_NO_CLUSTER", "1")
    b, xb, _ = run()
    monkeypatch.delenv("FNSSL_BWD_NO_CLUSTER")
    monkeypatch.setenv("FNSSL_CLUSTER_TEST_STALL", "3")
    monkeypatch.setenv("FNSSL_CLUSTER_SPIN_LIMIT", "2000")
    a, xa, word = run()
    assert word != 0 and (word >> 16) in (5, 6) and (word & 0xffff) == 0, hex(word)
    assert torch.equal(a, b) and torch.equal(xa, xb)
    monkeypatch.delenv("FNSSL_CLUSTER_TEST_STALL")
    a, xa, word = run()
    assert word == 0 and torch.equal(a, b) and torch.equal(xa, xb)


@pytest.mark.parametrize("online", [True, False])
def test_c_abi_training_step_matches_python_engine(dev, online):
    """fnssl_train_backward / fnssl_train_step (the whole step behind the C ABI, weight gradients through rocBLAS called
    from the library, h_prev as a shifted view) against the Python-orchestrated TrainEngine on the same flat vectors:
    same masks, same kernels -> gradients agree to GEMM summation order, then the same Adam update."""
    from fnssl import train
    nb, npair, nf, nt = 2, 3, 16, 24
    x = to_dev(rs_randn(7001, (nb * npair, 4, nf, nt)), dev)
    gt = to_dev(np.tanh(rs_randn(7002, (nb, nt // 12, 2 * nf, npair))), dev)
    _, _, eng_py = _engine(dev, online, 61, seed=4, process_group=False)
    _, _, eng_c = _engine(dev, online, 61, seed=4, process_group=False)
    loss_py = eng_py.step(x, gt)
    g_py = eng_py.grad.clone()
    cs = train.CTrainStep(eng_c)
    # gradients only
    eng_c.grad.zero_(), eng_c.loss_dev.zero_()
    cs.backward(x, gt, eng_py.last_seed, pair0=0)
    scale = float(g_py.abs().max())
    assert float((eng_c.grad - g_py).abs().max()) <= 2e-5 * scale
    assert abs(float(eng_c.loss_dev.item()) - loss_py) <= 1e-6 * max(1.0, abs(loss_py))
    # chunked accumulation == whole batch (global pair index keys the masks)
    g_whole = eng_c.grad.clone()
    eng_c.grad.zero_(), eng_c.loss_dev.zero_()
    n_total = nb * npair * (nt // 12) * 2 * nf
    for u in range(nb):
        cs.backward(x[u * npair:(u + 1) * npair], gt[u:u + 1], eng_py.last_seed, pair0=u * npair, n_total=n_total)
    assert float((eng_c.grad - g_whole).abs().max()) <= 2e-5 * scale
    # the one-call step leaves the parameters the Python engine's step left
    loss_c = cs.step(x, gt, eng_py.last_seed)
    assert abs(loss_c - loss_py) <= 1e-6 * max(1.0, abs(loss_py))
    big = g_py.abs() > 1e-4 * scale
    assert float((eng_c.theta - eng_py.theta)[big].abs().max()) <= 2e-6
    assert float(eng_c.theta[0]) == 0.0


def test_config4_real_shard_chunked_whole_deterministic_and_sampled_utterance_vs_oracle(dev):
    """BASELINE config 4's REAL per-GPU shard: 32 two-mic utterances x 256 bins x 300 frames (the launch paths only
    this size reaches: 4-waves-per-group ring-free LSTM kernels, 64-way split weight-gradient products, the 132 GB plan).
      * deterministic: two fresh engines, same seed -> bit-identical gradients and loss;
      * chunked == whole on the gradient (8 pairs per chunk, and ONE pair per chunk — the small-batch launch path —
        which ties the full-shard kernels to the path the oracle checks below);
      * a sampled utterance (global pair index 17, so it draws the masks it has inside the batch): loss, prediction and
        every gradient tensor against oracle/train_ref.py (PyTorch CPU autograd restatement of training_step)."""
    import gc
    from fnssl import ops
    from oracle import train_ref as T
    nb, nf, nt, seed, u = 32, 256, 300, 424242, 17
    sig = rs_randn(9100, (nb, 256 * (nt + 1), 2), 0.1)
    gt = np.tanh(rs_randn(9101, (nb, nt // 12, 2 * nf, 1)))
    xd = ops.preprocess(to_dev(sig, dev), "MM", layout=1)                 # [32, 4, 256, 300]
    gd = to_dev(gt, dev)

    def run(**kw):
        sd, net, eng = _engine(dev, True, 91, seed=1, process_group=False, **kw)
        eng.force_seed = seed
        loss = eng.step(xd, gd)
        g = eng.grad.detach().clone()
        del eng, net
        gc.collect()
        ops.release_workspaces()
        torch.cuda.empty_cache()
        return sd, loss, g

    sd, loss_w, g_w = run()
    _, loss_w2, g_w2 = run()
    assert loss_w2 == loss_w and torch.equal(g_w, g_w2), "the whole-shard step is not deterministic"
    scale = float(g_w.abs().max())
    assert scale > 0 and np.isfinite(loss_w)
    for cp in (8, 1):
        _, loss_c, g_c = run(chunk_pairs=cp)
        assert abs(loss_c - loss_w) <= 1e-5 * abs(loss_w), (cp, loss_c, loss_w)
        assert float((g_c - g_w).abs().max()) <= 2e-5 * scale, ("chunk_pairs", cp, float((g_c - g_w).abs().max()), scale)
    del g_w2, g_c
    # the sampled utterance alone, with the masks of its place in the batch
    sd, net, eng = _engine(dev, True, 91, seed=1, process_group=False)
    eng.force_seed = seed
    loss_u = eng.step(xd[u:u + 1], gd[u:u + 1], pair_offset=u)
    got = {k: v.cpu().numpy() for k, v in eng.gradients().items()}
    x_u = xd[u:u + 1].cpu().numpy()
    want_loss, grads, _, _, _ = T.train_step(sd, x_u, gt[u:u + 1], seed, 256, True, b0=u)
    assert abs(loss_u - want_loss) <= 2e-5 * abs(want_loss), (loss_u, want_loss)
    for k in grads:
        rel_close(got[k], grads[k], 1e-3, "grad " + k + " (utterance 17 of the config-4 shard)")


@pytest.mark.parametrize("H,ndir,c0,c2,nseq,nsteps", [(128, 2, 4, 0, 7, 9), (128, 2, 256, 0, 40, 33), (256, 1, 256, 4, 24, 50),
                                                      (256, 1, 256, 0, 3, 300), (128, 2, 132, 8, 1, 17)])
def test_weight_grads_kernel_vs_fp64_products(dev, H, ndir, c0, c2, nseq, nsteps):
    """fnssl_lstm_weight_grads (csrc/wgrad.hip: split-K fp32-MFMA product, h_prev by index shift, both directions, bias
    sums, deterministic reduction) against the float64 products it stands for: dW_ih = dA^T [x0 | x2],
    dW_hh = dA^T h_prev (one-step shift inside each sequence, zero at the boundary), db = sum dA; accumulation (+=);
    row counts that are not multiples of the 16-row stage; strided operands; run-to-run bit-stability."""
    from fnssl import ops
    rows = nseq * nsteps
    rng = np.random.RandomState(H + c0 + nseq)
    da = rng.standard_normal((rows, ndir * 4 * H + 8)).astype(np.float32)[:, :ndir * 4 * H]      # row stride > width
    x0 = rng.standard_normal((rows, c0)).astype(np.float32)
    x2 = rng.standard_normal((rows, c2)).astype(np.float32) if c2 else None
    h = rng.standard_normal((rows, ndir * H)).astype(np.float32)
    dad = to_dev(np.ascontiguousarray(rng.standard_normal((rows, ndir * 4 * H + 8)).astype(np.float32)), dev)[:, :ndir * 4 * H]
    dad.copy_(to_dev(da, dev))
    assert dad.stride(0) == ndir * 4 * H + 8
    x0d, hd = to_dev(x0, dev), to_dev(h, dev)
    x2d = to_dev(x2, dev) if c2 else None
    init = 0.5

    def run():
        g = {k: [torch.full(s, init, device=dev) for _ in range(ndir)] for k, s in
             (("wih", (4 * H, c0 + c2)), ("whh", (4 * H, H)), ("bih", (4 * H,)), ("bhh", (4 * H,)))}
        ops.lstm_weight_grads(dad, x0d, x2d, hd, H, ndir, nsteps, g["wih"], g["whh"], g["bih"], g["bhh"])
        return g

    g1, g2 = run(), run()
    x = np.concatenate([x0] + ([x2] if c2 else []), axis=1).astype(np.float64)
    h3 = h.reshape(nseq, nsteps, ndir * H).astype(np.float64)
    for d in range(ndir):
        a = da[:, d * 4 * H:(d + 1) * 4 * H].astype(np.float64)
        hp = np.zeros((nseq, nsteps, H))
        if d == 0:
            hp[:, 1:] = h3[:, :-1, :H]
        else:
            hp[:, :-1] = h3[:, 1:, H:]
        want = {"wih": a.T @ x, "whh": a.T @ hp.reshape(rows, H), "bih": a.sum(0), "bhh": a.sum(0)}
        for k, w in want.items():
            got = g1[k][d].cpu().numpy().astype(np.float64) - init
            scale = np.abs(w).max() + 1e-30
            assert np.abs(got - w).max() <= 2e-5 * scale + 1e-5, (k, d, np.abs(got - w).max(), scale)
            assert torch.equal(g1[k][d], g2[k][d]), "weight-gradient kernel is not deterministic"
    with pytest.raises(RuntimeError):
        ops.lstm_weight_grads(dad, x0d, x2d, hd[:, :-4], H, ndir, nsteps, g1["wih"], g1["whh"], g1["bih"], g1["bhh"])
