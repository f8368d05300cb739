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
