"""CPU baseline: the reference's forward path restated with the SAME PyTorch CPU
ops the reference calls (torch.stft, nn.LSTM -> oneDNN, nn.Linear, AvgPool2d).
TEST / MEASUREMENT INFRASTRUCTURE ONLY — imported by tests and by bench.py's
``cpu_baseline`` leg, never by the product path.

This is what a user of the reference gets on the host CPU, so it is the honest
number to time beside the GPU path (``cpu_baseline.kind = "port"``: the
reference itself cannot travel to the GPU box).  It is pinned to the same golden
vectors as the numpy oracle (tests/test_oracle_golden.py::test_torch_ref_*).

Reference lines restated: FN-SSL/Model.py:6-90 (FNblock, FN_SSL),
FN-SSL/Module.py:48-68 (STFT), :383-404 (AddChToBatch), FN-SSL/utils.py:9-55
(forgetting_norm), FN-SSL/Lightning/main.py:184-189,200-225 (predict_step,
data_preprocess).
"""
import torch
import torch.nn as nn


class RefFNblock(nn.Module):
    def __init__(self, input_size, hidden_size=256, is_online=False, is_first=False):
        super().__init__()
        fh = hidden_size // 2
        nh = hidden_size if is_online else hidden_size // 2
        self.is_first = is_first
        self.fullLstm = nn.LSTM(input_size, fh, batch_first=True, bidirectional=True)
        self.narrLstm = nn.LSTM(2 * fh + (input_size if is_first else 0), nh, batch_first=True,
                                bidirectional=not is_online)

    def forward(self, x, fb_skip=None):
        nb, nt, nf, _ = x.shape
        nb_skip = x.permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        x = x.reshape(nb * nt, nf, -1)
        if not self.is_first:
            x = x + fb_skip
        x, _ = self.fullLstm(x)
        fb_skip = x
        x = x.view(nb, nt, nf, -1).permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        x = torch.cat((x, nb_skip), dim=-1) if self.is_first else x + nb_skip
        x, _ = self.narrLstm(x)
        return x.view(nb, nf, nt, -1).permute(0, 2, 1, 3), fb_skip


class RefFNSSL(nn.Module):
    def __init__(self, input_size=4, is_online=True):
        super().__init__()
        self.block_1 = RefFNblock(input_size, 256, is_online, True)
        self.block_2 = RefFNblock(256, 256, is_online, False)
        self.block_3 = RefFNblock(256, 256, is_online, False)
        self.emb2ipd = nn.Linear(256, 2)
        self.pooling = nn.AvgPool2d(kernel_size=(12, 1))

    def forward(self, x):
        x = x.permute(0, 3, 2, 1)
        nb, nt, nf, _ = x.shape
        x, fb = self.block_1(x)
        x, fb = self.block_2(x, fb)
        x, fb = self.block_3(x, fb)
        x = x.permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        ipd = torch.tanh(self.emb2ipd(self.pooling(x)))
        nt2 = ipd.shape[1]
        ipd = ipd.view(nb, nf, nt2, -1).permute(0, 2, 1, 3)
        return torch.cat((ipd[..., 0], ipd[..., 1]), dim=2)


def forgetting_norm(mag, sample_length=298):
    B, Cn, Fq, T = mag.shape
    m = mag.reshape(B, Cn * Fq, T)
    alpha = (sample_length - 1) / (sample_length + 1)
    mu = 0
    outs = []
    for t in range(T):
        mean_t = torch.mean(m[:, :, t], dim=1).reshape(B, 1)
        if t < sample_length:
            alp = torch.min(torch.tensor([(t - 1) / (t + 1), alpha]))
            mu = alp * mu + (1 - alp) * mean_t
        else:
            mu = alpha * mu + (1 - alpha) * mean_t
        outs.append(mu)
    return torch.stack(outs, dim=-1).reshape(B, 1, 1, T)


def data_preprocess(sig, ch_mode="MM", eps=1e-6):
    """sig [nb, ns, nch] -> [nb*np, 4, 256, nt]."""
    nb, ns, nch = sig.shape
    win = torch.hann_window(512)
    spec = torch.stack([torch.stft(sig[:, :, c], n_fft=512, hop_length=256, win_length=512, window=win,
                                   center=False, normalized=False, return_complex=True) for c in range(nch)],
                       dim=1)                                              # [nb, nch, 257, nt]
    pairs = ([(0, j) for j in range(1, nch)] if ch_mode == "M"
             else [(i, j) for i in range(nch - 1) for j in range(i + 1, nch)])
    idx = torch.tensor(pairs)
    reb = spec[:, idx].reshape((nb * len(pairs), 2) + tuple(spec.shape[2:]))
    mu = forgetting_norm(torch.abs(reb))
    x = torch.cat((torch.real(reb) / (mu + eps), torch.imag(reb) / (mu + eps)), dim=1)
    return x[:, :, 1:257, :]


def build(state: dict, is_online: bool = True) -> RefFNSSL:
    net = RefFNSSL(is_online=is_online).eval()
    missing = net.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()
                                   if not k.startswith("ipd2doa")}, strict=True)
    del missing
    return net


@torch.no_grad()
def predict_step(net: RefFNSSL, batch, ch_mode="MM"):
    """batch [nb, nch, ns] -> [nb*np, nt//12, 512]."""
    return net(data_preprocess(batch.permute(0, 2, 1), ch_mode))


# ------------------------------------------------------------------------------------------------------------ #
# IPDnet (fixed array) and IPDnet2 (OnlineSpatialNet) on PyTorch CPU ops — the multi-threaded cpu_baseline of
# BASELINE configs 3 and 5 ("stock torch.nn-based module, torch.set_num_threads(all cores)").
# Pinned to the same reference-generated fixtures as the numpy oracles (tests/test_oracle_golden.py,
# tests/test_oracle_ipdnet2.py).
# ------------------------------------------------------------------------------------------------------------ #
class RefIPDnetBlock(nn.Module):
    """IPDnet/FixedAarryIPDnet.py:11-40 (eval mode: the dropouts are identities)."""

    def __init__(self, input_size, hidden_size, add_skip_dim, is_online, is_first):
        super().__init__()
        fh = hidden_size // 2
        nh = hidden_size if is_online else hidden_size // 2
        self.fullLstm = nn.LSTM(input_size + (0 if is_first else add_skip_dim), fh, batch_first=True, bidirectional=True)
        self.narrLstm = nn.LSTM(2 * fh + add_skip_dim, nh, batch_first=True, bidirectional=not is_online)

    def forward(self, x, fb_skip, nb_skip):
        nb, nt, nf, _ = x.shape
        x, _ = self.fullLstm(x.reshape(nb * nt, nf, -1))
        x = torch.cat((x, fb_skip), dim=-1)
        x = x.view(nb, nt, nf, -1).permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        x, _ = self.narrLstm(x)
        x = torch.cat((x, nb_skip), dim=-1)
        return x.view(nb, nf, nt, -1).permute(0, 2, 1, 3)


class RefCausCnn(nn.Module):
    """FixedAarryIPDnet.py:47-73."""

    def __init__(self, inp_dim, out_dim, hid=128):
        super().__init__()
        self.conv1 = nn.Conv2d(inp_dim, hid, 3, padding=(1, 2), bias=False)
        self.conv2 = nn.Conv2d(hid, hid, 3, padding=(1, 2), bias=False)
        self.conv3 = nn.Conv2d(hid, out_dim, 3, padding=(1, 2), bias=False)

    def forward(self, x):
        x = torch.nn.functional.avg_pool2d(torch.relu(self.conv1(x))[:, :, :, :-2], (1, 3))
        x = torch.nn.functional.avg_pool2d(torch.relu(self.conv2(x))[:, :, :, :-2], (1, 4))
        return torch.tanh(self.conv3(x)[:, :, :, :-2])


class RefIPDnet(nn.Module):
    """FixedAarryIPDnet.py:80-120 (whole-signal path)."""

    def __init__(self, input_size=4, hidden_size=128, max_track=2, is_online=True):
        super().__init__()
        self.block_1 = RefIPDnetBlock(input_size, hidden_size, input_size, is_online, True)
        self.block_2 = RefIPDnetBlock(hidden_size, hidden_size, input_size, is_online, False)
        self.conv = RefCausCnn(hidden_size + input_size, 2 * (input_size // 2 - 1) * max_track)

    def forward(self, x):
        x = x.permute(0, 3, 2, 1)
        nb, nt, nf, _ = x.shape
        fb_skip = x.reshape(nb * nt, nf, -1)
        nb_skip = x.permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        x = self.block_1(x, fb_skip, nb_skip)
        x = self.block_2(x, fb_skip, nb_skip)
        nt2 = nt // 12
        x = self.conv(x.permute(0, 3, 2, 1)).permute(0, 3, 2, 1)
        x = x.reshape(nb, nt2, nf, 2, -1).permute(0, 1, 3, 2, 4)
        return x.reshape(nb, nt2, 2, nf * 2, -1).permute(0, 1, 3, 4, 2)


def build_ipdnet(state: dict, input_size, hidden_size, max_track=2, is_online=True) -> RefIPDnet:
    net = RefIPDnet(input_size, hidden_size, max_track, is_online).eval()
    net.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()}, strict=True)
    return net


def array_preprocess(sig, sample_length=280, hop=256, center=False, eps=1e-6):
    """sig [nb, ns, nch] -> [nb, 2 nch, 256, nt] with torch.stft (IPDnet/runIPDnetOn.py:240-254; hop 320, centred,
    sample_length 249: IPDnet2/run_IPDnet2.py:277-288)."""
    win = torch.hann_window(512)
    spec = torch.stack([torch.stft(sig[:, :, c], n_fft=512, hop_length=hop, win_length=512, window=win, center=center,
                                   normalized=False, return_complex=True) for c in range(sig.shape[2])], dim=1)
    mu = forgetting_norm(torch.abs(spec), sample_length)
    x = torch.cat((torch.real(spec) / (mu + eps), torch.imag(spec) / (mu + eps)), dim=1)
    return x[:, :, 1:257, :]


def _ln(x, w, b):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _mamba_torch(sd, p, x):
    """One Mamba block with torch CPU ops (same restatement as oracle/ipdnet2_oracle.py::mamba; PARITY UNPINNED:
    the reference's own block is mamba_ssm's CUDA kernel, absent here).  x [S, T, D] -> [S, T, D]."""
    F_ = torch.nn.functional
    g = lambda k: sd[p + k]   # noqa: E731
    E = g("in_proj.weight").shape[0] // 2
    xz = x @ g("in_proj.weight").T
    xi, z = xz[..., :E], xz[..., E:]
    K = g("conv1d.weight").shape[2]
    u = F_.conv1d(xi.transpose(1, 2), g("conv1d.weight"), g("conv1d.bias"), padding=K - 1, groups=E)[..., :x.shape[1]]
    u = F_.silu(u).transpose(1, 2)
    N = g("A_log").shape[1]
    R = g("x_proj.weight").shape[0] - 2 * N
    dbl = u @ g("x_proj.weight").T
    dt = F_.softplus(dbl[..., :R] @ g("dt_proj.weight").T + g("dt_proj.bias"))
    Bm, Cm = dbl[..., R:R + N], dbl[..., R + N:]
    A = -torch.exp(g("A_log"))
    h = torch.zeros((x.shape[0], E, N))
    ys = []
    for t in range(x.shape[1]):
        h = torch.exp(dt[:, t, :, None] * A) * h + (dt[:, t, :, None] * Bm[:, t, None, :]) * u[:, t, :, None]
        ys.append((h * Cm[:, t, None, :]).sum(-1) + g("D") * u[:, t])
    y = torch.stack(ys, 1) * F_.silu(z)
    return y @ g("out_proj.weight").T


@torch.no_grad()
def ipdnet2_forward(state: dict, x, time_ratio=5, ratio=16):
    """OnlineSpatialNet.forward (IPDnet2/IPDnet2.py:331-368, shipped configuration) on torch CPU ops.
    x [B, C, F, T] -> [B, T // 5, 2F, 4, 2]."""
    F_ = torch.nn.functional
    sd = {k: torch.as_tensor(v) for k, v in state.items()}
    x = x.permute(0, 2, 3, 1)
    B, F, T, H0 = x.shape
    xe = F_.pad(x.reshape(B * F, T, H0).permute(0, 2, 1), (4, 0))
    x = F_.conv1d(xe, sd["encoder.weight"], sd["encoder.bias"]).permute(0, 2, 1).reshape(B, F, T, -1)
    nl = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))

    def fconv(p, x):
        Bq, Fq, Tq, Hq = x.shape
        y = _ln(x, sd[p + ".0.weight"], sd[p + ".0.bias"]).permute(0, 2, 3, 1).reshape(Bq * Tq, Hq, Fq)
        y = F_.conv1d(y, sd[p + ".1.weight"], sd[p + ".1.bias"], padding="same", groups=8)
        y = F_.prelu(y, sd[p + ".2.weight"])
        return y.reshape(Bq, Tq, Hq, Fq).permute(0, 3, 1, 2)

    def full(p, x):
        Bq, Fq, Tq, Hq = x.shape
        y = _ln(x, sd[p + "norm_full.weight"], sd[p + "norm_full.bias"])
        s = F_.silu(y @ sd[p + "squeeze.0.weight"][:, :, 0].T + sd[p + "squeeze.0.bias"])       # [B, F, T, 8]
        s = torch.einsum("bftq,gf->bgtq", s, sd[p + "full.weight"]) + sd[p + "full.bias"][None, :, None, None]
        return F_.silu(s @ sd[p + "unsqueeze.0.weight"][:, :, 0].T + sd[p + "unsqueeze.0.bias"])

    def mam(pn, pm, x):
        Bq, Fq, Tq, Hq = x.shape
        y = _ln(x, sd[pn + ".weight"], sd[pn + ".bias"]).reshape(Bq * Fq, Tq, Hq)
        return _mamba_torch(sd, pm + ".", y).reshape(Bq, Fq, Tq, Hq)

    for l in range(nl):
        p = "layers.%d." % l
        x = x + fconv(p + "fconv1", x)
        if l == 0:
            x = x.reshape(B, x.shape[1] // 2, 2, T, -1).mean(2)
        x = x + full(p, x)
        x = x + fconv(p + "fconv2", x)
        if l == 0:
            x = x.reshape(B, x.shape[1] // 8, 8, T, -1).mean(2)
        x = x + mam(p + "norm_mhsa", p + "mhsa", x)
        x = x + mam(p + "norm_tconvffn", p + "tconvffn", x)
        if l == 0:
            T2 = T // time_ratio
            x = x[:, :, :T2 * time_ratio].reshape(B, x.shape[1], T2, time_ratio, -1).mean(3)
    Bq, Fc, T2, Hq = x.shape
    w, b = sd["freq_inverse.trans2.weight"][:, :, 0], sd["freq_inverse.trans2.bias"]
    do = sd["decoder.weight"].shape[0]
    y = x @ w.T + b                                                       # [B, Fc, T2, ratio * do]; o = c * ratio + r
    y = torch.tanh(y.reshape(Bq, Fc, T2, do, ratio).permute(0, 1, 4, 2, 3).reshape(Bq, Fc * ratio, T2, do))
    y = y @ sd["decoder.weight"].T + sd["decoder.bias"]
    y = y.permute(0, 2, 1, 3).reshape(Bq, T2, F, 2, -1).permute(0, 1, 3, 2, 4)
    return y.reshape(Bq, T2, 2, F * 2, -1).permute(0, 1, 3, 4, 2).contiguous()
