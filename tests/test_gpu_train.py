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
