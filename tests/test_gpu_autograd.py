"""GPU parity tests of the AUTOGRAD route (SURVEY.md §8b "Callers": training_step needs autograd through forward):
``Model.FN_SSL`` / ``Model.FNblock`` in ``train()`` mode return tensors with a ``grad_fn`` whose backward runs the HIP
BPTT / weight-gradient kernels (fnssl/autograd.py), so the reference's loops

    pred = self(in_batch); loss = cal_loss(pred, gt); loss.backward(); optimizer.step()
        FN-SSL/Lightning/main.py:149-157, FN-SSL/Learner.py:104-115

work as written.  Checked against the PyTorch CPU autograd oracle (oracle/train_ref.py), the real reference's golden
gradients (tests/golden/g13_train.npz) and the fused ``TrainEngine`` (same kernels, same numbers)."""
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


def _net(dev, online, wseed):
    import Model
    from fnssl import weights as W
    sd = W.make_fnssl_state(wseed, 4, 256, online)
    net = Model.FN_SSL(is_online=online)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return sd, net.to(dev)


def _ref_cal_loss(pred, gt):
    """main.py:191-198 with the ATen ops the reference calls (RemoveChFromBatch = a reshape)."""
    nb = gt.shape[0]
    reb = pred.reshape((nb, pred.shape[0] // nb) + tuple(pred.shape[1:])).permute(0, 2, 3, 1)
    return torch.nn.functional.mse_loss(reb.contiguous(), gt.contiguous())


@pytest.mark.parametrize("case", [0, 1])
def test_train_mode_forward_backward_adam_reproduce_the_reference_golden(dev, case):
    """net.train(); loss = mse(net(x), gt); loss.backward(); torch.optim.Adam.step() == G13 (loss, the real reference's
    gradient norms, the oracle's every gradient, the parameters after the step) to the engine test's tolerances."""
    from oracle import train_ref as T
    g = load_golden("g13_train")
    online, nb, npair, nf, nt, seed, wseed, xseed, gseed = [int(v) for v in g["c%d_cfg" % case]]
    sd, net = _net(dev, bool(online), wseed)
    x = rs_randn(xseed, (nb * npair, 4, nf, nt))
    gt = rs_randn(gseed, (nb, nt // 12, 2 * nf, npair), 0.5)
    net.train()
    import fnssl.train as tr
    orig = tr.layer_seed
    try:
        tr.layer_seed = lambda base, l: orig(seed, l)             # the golden's masks
        in_batch = to_dev(x, dev).requires_grad_()                # Learner.py:102
        pred = net(in_batch)
        assert pred.requires_grad and pred.grad_fn is not None
        loss = _ref_cal_loss(pred, to_dev(gt, dev))
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        opt.zero_grad()
        loss.backward()
    finally:
        tr.layer_seed = orig
    assert in_batch.grad is None                                  # documented: the feature gradient is not produced
    want_loss, grads, new_sd, _, want_pred = T.train_step(sd, x, gt, seed, 256, bool(online))
    assert_close(pred.detach().cpu().numpy(), want_pred, 1e-4, 1e-5, "train-mode forward (dropout on)")
    assert abs(float(loss.detach()) - want_loss) <= 1e-5 * abs(want_loss)
    assert abs(float(loss.detach()) - float(g["c%d_loss" % case])) <= 1e-5 * abs(want_loss)
    names = [str(s) for s in g["c%d_names" % case]]
    got = {k: p.grad for k, p in net.named_parameters()}
    assert names == list(got.keys())
    for i, k in enumerate(names):
        assert got[k] is not None and got[k].shape == net.state_dict()[k].shape, k
        rel_close(got[k].cpu().numpy(), grads[k], 5e-4, "grad " + k)
        a = got[k].cpu().numpy().astype(np.float64)
        want = g["c%d_gproj" % case][i]
        assert abs(np.sqrt((a * a).sum()) - want[0]) <= 5e-4 * want[0] + 1e-12, ("norm vs reference", k)
    opt.step()
    for k in names:
        d_got = net.state_dict()[k].cpu().numpy() - sd[k]
        d_want = new_sd[k] - sd[k]
        big = np.abs(grads[k]) > 1e-3 * np.abs(grads[k]).max()
        assert np.abs(d_got - d_want)[big].max() <= 2e-5, ("adam update", k)
    # the inference entry sees the stepped weights (packed-weight caches are keyed on the parameters' versions)
    net.eval()
    with torch.no_grad():
        y = net(to_dev(x, dev))
    assert y.grad_fn is None and torch.isfinite(y).all()


def test_autograd_route_equals_fused_engine(dev):
    """Same kernels, same masks -> the parameter gradients of loss.backward() are the engine's, bit for bit, when the loss
    gradient comes from the same kernel (predict_step._MSELoss); two steps with a torch optimizer track the engine's Adam."""
    import predict_step as ps
    from fnssl import autograd as ag
    from fnssl import train
    nb, npair, nf, nt = 2, 3, 8, 24
    x = rs_randn(61, (nb * npair, 4, nf, nt))
    gt = rs_randn(62, (nb, nt // 12, 2 * nf, npair), 0.5)
    sd, net_a = _net(dev, True, 88)
    _, net_e = _net(dev, True, 88)
    eng = train.TrainEngine(net_e, seed=7, process_group=False)
    net_a.train()
    net_a.dropout_seed = 7
    opt = torch.optim.Adam(net_a.parameters(), lr=1e-3)
    for step in (1, 2):
        le = eng.step(to_dev(x, dev), to_dev(gt, dev))
        opt.zero_grad()
        loss = ps._MSELoss.apply(net_a(to_dev(x, dev)), to_dev(gt, dev))
        loss.backward()
        assert net_a.last_dropout_base == ag.base_seed(7, step) == eng.last_seed
        assert abs(float(loss.detach()) - le) <= 1e-6 * abs(le)
        ge = eng.gradients()
        for k, p in net_a.named_parameters():
            if step == 1:
                assert torch.equal(p.grad, ge[k]), k
            else:
                rel_close(p.grad.cpu().numpy(), ge[k].cpu().numpy(), 1e-3, "step-2 grad " + k)
        opt.step()
        for k, p in net_a.named_parameters():
            # Adam moves a weight by ~lr * g / (|g| + eps): compare where the gradient is not down at eps
            big = ge[k].abs() > 1e-3 * ge[k].abs().max()
            assert (p.detach() - net_e.state_dict()[k])[big].abs().max() <= 2e-6, (step, k)


def test_learner_style_loop_and_gradient_accumulation(dev):
    """Learner.train_epoch's loop (Learner.py:95-115): zero_grad / forward / loss / backward / step; the loss falls on a
    repeated batch.  Two backward() calls without zero_grad accumulate (.grad += ), as autograd does for nn.LSTM."""
    nb, npair, nf, nt = 2, 1, 16, 24
    x = to_dev(rs_randn(71, (nb * npair, 4, nf, nt)), dev)
    gt = to_dev(np.tanh(rs_randn(72, (nb, nt // 12, 2 * nf, npair))), dev)
    _, net = _net(dev, True, 5)
    net.train()
    net.dropout_seed = 1
    optimizer = torch.optim.Adam(net.parameters(), lr=1e-3)
    optimizer.zero_grad()
    losses = []
    for _ in range(6):
        in_batch = x.clone().requires_grad_()
        pred_batch = net(in_batch)
        loss_batch = _ref_cal_loss(pred_batch, gt)
        loss_batch.backward()
        optimizer.step()
        optimizer.zero_grad()
        losses.append(float(loss_batch))
    assert losses[-1] < losses[0], losses
    # accumulation
    net.force_dropout_base = 1234
    net.zero_grad()
    _ref_cal_loss(net(x), gt).backward()
    g1 = {k: p.grad.clone() for k, p in net.named_parameters()}
    _ref_cal_loss(net(x), gt).backward()
    for k, p in net.named_parameters():
        assert_close(p.grad.cpu().numpy(), 2 * g1[k].cpu().numpy(), 1e-6, 1e-12, "accumulated " + k)
    # a frozen parameter gets no gradient, the others are unaffected
    net.zero_grad()
    net.emb2ipd.bias.requires_grad_(False)
    _ref_cal_loss(net(x), gt).backward()
    assert net.emb2ipd.bias.grad is None
    assert torch.equal(net.block_2.fullLstm.weight_hh_l0.grad, g1["block_2.fullLstm.weight_hh_l0"])
    # backward twice through one forward: a clear error, not stale memory
    net.emb2ipd.bias.requires_grad_(True)
    y = net(x)
    y.sum().backward()
    with pytest.raises(RuntimeError):
        y.sum().backward()
    # eval mode: the forward-only kernels, no graph
    net.eval()
    assert net(x).grad_fn is None


@pytest.mark.parametrize("first,online", [(False, True), (True, True), (False, False)])
def test_fnblock_train_mode_matches_autograd_oracle(dev, first, online):
    """FNblock.forward in train mode on its own (Model.py:31-50): the three outputs with dropout active and the
    gradients w.r.t. x, fb_skip and the block's parameters, against oracle.train_ref.TrainFNblock under CPU autograd."""
    import Model
    from fnssl import autograd as ag
    from fnssl import train
    from fnssl import weights as W
    from oracle import train_ref as T
    nb, nt, nf = 2, 5, 6
    cin = 4 if first else 256
    blk = Model.FNblock(input_size=cin, is_online=online, is_first=first)
    sd = W.make_state([(k, tuple(v.shape)) for k, v in blk.state_dict().items()], seed=321 + cin)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    blk = blk.to(dev).train()
    blk.force_dropout_base, blk.dropout_layer, blk.pair_offset = 99, 2, 3
    nh = 256 if online else 256
    x = rs_randn(1, (nb, nt, nf, cin), 0.7)
    fb = rs_randn(2, (nb * nt, nf, 256), 0.7)
    gs = [rs_randn(3, (nb, nt, nf, nh)), rs_randn(4, (nb * nt, nf, 256)), rs_randn(5, (nb * nf, nt, nh))]
    # ---- oracle
    ob = T.TrainFNblock(cin, 256, online, first)
    ob.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    xo = torch.from_numpy(x).requires_grad_(True)
    fo = torch.from_numpy(fb).requires_grad_(True)
    m_full = torch.from_numpy(T.dropout_scale(T.layer_seed(99, 2), (nb, nt, nf, 256), b0=3))
    m_narr = torch.from_numpy(T.dropout_scale(T.layer_seed(99, 3), (nb, nt, nf, nh), b0=3))
    yo, fbo = ob(xo, None if first else fo, m_full, m_narr)
    # nb_skip = the narrow-band output before dropout: recompute it through the oracle with an all-ones mask
    yo1, _ = ob(xo, None if first else fo, m_full, torch.ones_like(m_narr))
    nbo = yo1.permute(0, 2, 1, 3).reshape(nb * nf, nt, nh)
    (yo * torch.from_numpy(gs[0])).sum().backward(retain_graph=True)
    (fbo * torch.from_numpy(gs[1])).sum().backward(retain_graph=True)
    (nbo * torch.from_numpy(gs[2])).sum().backward()
    # ---- HIP
    xd = to_dev(x, dev).requires_grad_(True)
    fd = to_dev(fb, dev).requires_grad_(True)
    y, fbs, nbs = blk(xd, None, None if first else fd)
    assert y.shape == (nb, nt, nf, nh) and fbs.shape == (nb * nt, nf, 256) and nbs.shape == (nb * nf, nt, nh)
    assert_close(y.detach().cpu().numpy(), yo.detach().numpy(), 1e-4, 1e-5, "x out (dropout_narr applied)")
    assert_close(fbs.detach().cpu().numpy(), fbo.detach().numpy(), 1e-4, 1e-5, "fb_skip (pre-dropout)")
    assert_close(nbs.detach().cpu().numpy(), nbo.detach().numpy(), 1e-4, 1e-5, "nb_skip (pre-dropout)")
    ((y * to_dev(gs[0], dev)).sum() + (fbs * to_dev(gs[1], dev)).sum() + (nbs * to_dev(gs[2], dev)).sum()).backward()
    for k, p in blk.named_parameters():
        rel_close(p.grad.cpu().numpy(), dict(ob.named_parameters())[k].grad.numpy(), 5e-4, "grad " + k)
    if first:
        assert xd.grad is None
    else:
        rel_close(xd.grad.cpu().numpy(), xo.grad.numpy(), 5e-4, "grad x")
        rel_close(fd.grad.cpu().numpy(), fo.grad.numpy(), 5e-4, "grad fb_skip")
    assert ag.base_seed(0, 1) == 8191 and train.layer_seed(99, 2) == T.layer_seed(99, 2)


def test_mymodel_automatic_optimization_route(dev):
    """predict_step.MyModel(fused_engine=False): training_step returns a loss WITH a graph (what Lightning's automatic
    optimisation calls backward() on), configure_optimizers returns the reference's Adam + ExponentialLR dictionary
    (main.py:269-279); a few steps lower the loss and the predict entry sees the new weights."""
    import predict_step as ps
    from fnssl import weights as W
    from oracle import train_ref as T
    m = ps.MyModel(device="cuda", fused_engine=False)
    m.arch.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in W.make_fnssl_state(5).items()})
    m.to(dev)
    sig = to_dev(rs_randn(41, (2, 512 + 23 * 256, 2), 0.1), dev)
    gt = to_dev(np.full((2, 2, 512, 1), 0.5, dtype=np.float32) + rs_randn(42, (2, 2, 512, 1), 0.05), dev)
    cfg = m.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    assert isinstance(opt, torch.optim.Adam) and isinstance(sched, torch.optim.lr_scheduler.ExponentialLR)
    m.eval()
    pred0 = m.predict_step(sig.permute(0, 2, 1))
    l0 = float(m.cal_loss(pred0, {"ipd": gt}))
    assert abs(l0 - float(T.cal_loss(pred0.cpu(), gt.cpu()))) <= 1e-6 * l0
    m.train()
    for _ in range(4):
        out = m.training_step((sig, {"ipd": gt}), 0)
        assert out["loss"].requires_grad and out["loss"].shape == ()
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
    sched.step()
    assert abs(opt.param_groups[0]["lr"] - 0.001 * 0.8988) < 1e-12
    m.eval()
    pred1 = m.predict_step(sig.permute(0, 2, 1))
    assert float(m.cal_loss(pred1, {"ipd": gt})) < l0
    # cal_loss's own gradient (one HIP kernel) against ATen's
    p = pred1.clone().requires_grad_()
    m.cal_loss(p, {"ipd": gt}).backward()
    q = pred1.clone().requires_grad_()
    _ref_cal_loss(q, gt).backward()
    assert_close(p.grad.cpu().numpy(), q.grad.cpu().numpy(), 1e-6, 1e-12, "d loss / d pred")
