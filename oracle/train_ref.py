"""Training-step oracle (BASELINE config 4): the reference's training forward
(dropout active), loss, backward and Adam update restated with PyTorch CPU autograd.
TEST / MEASUREMENT INFRASTRUCTURE ONLY — imported by tests and tools, never by the product path.

Reference lines restated: FN-SSL/Model.py:31-50 (FNblock.forward with dropout_full / dropout_narr),
:72-90 (FN_SSL.forward), FN-SSL/Lightning/main.py:149-157 (training_step), :191-198 (cal_loss, MSE on the
re-batched prediction), :269-271 (Adam, lr 1e-3), FN-SSL/Module.py:406-421 (RemoveChFromBatch).

Dropout: the reference draws Bernoulli masks from torch's RNG, which no other implementation can
reproduce.  Here (and in the HIP path) the keep-mask of layer l is a pure function of
(seed, l, logical element index) — ``dropout_scale`` below is the numpy restatement of the device hash
(csrc/train.hip: keep_scale) — and the golden vectors were produced by the REAL reference with exactly these
masks injected in place of nn.Dropout's sampling (tests/golden/make_golden_train.py).
"""
import numpy as np
import torch
import torch.nn as nn

from .torch_ref import RefFNSSL  # noqa: F401  (same module structure)

M32 = np.uint64(0xFFFFFFFF)
KEEP_THRESHOLD = 13421773          # round(0.8 * 2**24): keep probability 0.8 = 1 - p, p = 0.2 (Model.py:9)
KEEP_SCALE = np.float32(1.25)      # 1 / (1 - p)


def _fmix32(h):
    h = h & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def layer_seed(seed: int, layer: int) -> int:
    """32-bit seed of dropout layer ``layer`` (0..5 = block_1.dropout_full, block_1.dropout_narr, ...)."""
    return int(_fmix32(np.uint64((seed + 0x9E3779B9 * (layer + 1)) & 0xFFFFFFFF)))


def dropout_scale(seed32: int, shape, b0: int = 0) -> np.ndarray:
    """Keep-scale tensor {0, 1.25} for a logical [nb, nt, nf, C] activation whose first utterance-pair is
    row ``b0`` of the rank's batch.  Element index = ((b*nt + t)*nf + f)*C + c."""
    nb, nt, nf, c = shape
    idx = (np.arange(nb * nt * nf * c, dtype=np.uint64) + np.uint64(b0 * nt * nf * c))
    lo, hi = idx & M32, idx >> np.uint64(32)
    h = _fmix32(((lo * np.uint64(0xCC9E2D51)) & M32) ^ np.uint64(seed32))
    h = _fmix32((h + ((hi * np.uint64(0x1B873593)) & M32) + np.uint64(seed32) * np.uint64(0x85EBCA6B)) & M32)
    keep = (h >> np.uint64(8)) < np.uint64(KEEP_THRESHOLD)
    return (keep.astype(np.float32) * KEEP_SCALE).reshape(shape)


class TrainFNblock(nn.Module):
    def __init__(self, input_size, hidden_size, is_online, is_first):
        super().__init__()
        fh = hidden_size // 2
        nh = hidden_size if is_online else hidden_size // 2
        self.is_first = is_first
        self.fullLstm = nn.LSTM(input_size, fh, batch_first=True, bidirectional=True)
        self.narrLstm = nn.LSTM(2 * fh + (input_size if is_first else 0), nh, batch_first=True,
                                bidirectional=not is_online)

    def forward(self, x, fb_skip, m_full, m_narr):
        """Model.py:31-50 with the two dropouts as explicit scale tensors (logical [nb, nt, nf, C])."""
        nb, nt, nf, _ = x.shape
        nb_skip = x.permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        x = x.reshape(nb * nt, nf, -1)
        if not self.is_first:
            x = x + fb_skip
        x, _ = self.fullLstm(x)
        fb_skip = x
        x = x * m_full.reshape(nb * nt, nf, -1)                                   # dropout_full :40
        x = x.view(nb, nt, nf, -1).permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        x = torch.cat((x, nb_skip), dim=-1) if self.is_first else x + nb_skip
        x, _ = self.narrLstm(x)
        x = x.view(nb, nf, nt, -1).permute(0, 2, 1, 3)
        return x * m_narr, fb_skip                                                # dropout_narr :48


class TrainFNSSL(nn.Module):
    def __init__(self, input_size=4, hidden_size=256, is_online=True):
        super().__init__()
        self.block_1 = TrainFNblock(input_size, hidden_size, is_online, True)
        self.block_2 = TrainFNblock(hidden_size, hidden_size, is_online, False)
        self.block_3 = TrainFNblock(hidden_size, hidden_size, is_online, False)
        self.emb2ipd = nn.Linear(hidden_size, 2)
        self.pooling = nn.AvgPool2d(kernel_size=(12, 1))

    def forward(self, x, masks):
        x = x.permute(0, 3, 2, 1)
        nb, nt, nf, _ = x.shape
        fb = None
        for k, blk in enumerate((self.block_1, self.block_2, self.block_3)):
            x, fb = blk(x, fb, masks[2 * k], masks[2 * k + 1])
        x = x.permute(0, 2, 1, 3).reshape(nb * nf, nt, -1)
        ipd = torch.tanh(self.emb2ipd(self.pooling(x)))
        nt2 = ipd.shape[1]
        ipd = ipd.view(nb, nf, nt2, -1).permute(0, 2, 1, 3)
        return torch.cat((ipd[..., 0], ipd[..., 1]), dim=2)


def cal_loss(pred, gt_ipd):
    """main.py:191-198: pred [nb*np, nt2, 2nf], gt_ipd [nb, nt2, 2nf, np] -> scalar MSE."""
    nb = gt_ipd.shape[0]
    npair = pred.shape[0] // nb
    reb = pred.reshape((nb, npair) + tuple(pred.shape[1:])).permute(0, 2, 3, 1)     # RemoveChFromBatch + permute
    return torch.nn.functional.mse_loss(reb.contiguous(), gt_ipd.contiguous())


def make_masks(seed, nbp, nt, nf, hidden_size, is_online=True, b0=0):
    fh2 = 2 * (hidden_size // 2)
    nh = hidden_size if is_online else 2 * (hidden_size // 2)
    return [torch.from_numpy(dropout_scale(layer_seed(seed, l), (nbp, nt, nf, fh2 if l % 2 == 0 else nh), b0))
            for l in range(6)]


def train_step(sd, x, gt_ipd, seed, hidden_size=256, is_online=True, lr=1e-3, adam_state=None, step=1,
               grad_divisor=1.0, b0=0):
    """One training step on CPU.  sd: numpy state dict (reference names); x [nb*np, 4, nf, nt] features;
    gt_ipd [nb, nt2, 2nf, np]; ``b0``: index of x's first pair in the GLOBAL batch (keys the dropout masks, so a
    slice of a larger batch draws that batch's masks).  Returns (loss, grads dict, new params dict, new adam_state)."""
    net = TrainFNSSL(4, hidden_size, is_online)
    net.load_state_dict({k: torch.from_numpy(np.array(v, dtype=np.float32)) for k, v in sd.items()})
    xt = torch.from_numpy(np.asarray(x, dtype=np.float32))
    nbp, _, nf, nt = xt.shape
    masks = make_masks(seed, nbp, nt, nf, hidden_size, is_online, b0)
    pred = net(xt, masks)
    loss = cal_loss(pred, torch.from_numpy(np.asarray(gt_ipd, dtype=np.float32)))
    loss.backward()
    names = [k for k, _ in net.named_parameters()]
    grads = {k: p.grad.numpy().copy() for k, p in net.named_parameters()}
    # torch.optim.Adam defaults (main.py:270): betas (0.9, 0.999), eps 1e-8, no weight decay
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    if adam_state is not None:
        for p, k in zip(net.parameters(), names):
            opt.state[p] = {"step": torch.tensor(float(step - 1)), "exp_avg": torch.from_numpy(adam_state[k][0].copy()),
                            "exp_avg_sq": torch.from_numpy(adam_state[k][1].copy())}
    if grad_divisor != 1.0:
        for p in net.parameters():
            p.grad.div_(grad_divisor)
    opt.step()
    new_sd = {k: p.detach().numpy().copy() for k, p in net.named_parameters()}
    new_state = {k: (opt.state[p]["exp_avg"].numpy().copy(), opt.state[p]["exp_avg_sq"].numpy().copy())
                 for p, k in zip(net.parameters(), names)}
    return float(loss.detach()), grads, new_sd, new_state, pred.detach().numpy()
