"""Drop-in for the reference's ``Model.py`` (FN-SSL/Model.py, = FN-SSL/Lightning/Model.py):
same classes, constructor arguments, ``forward`` signatures, return shapes and
``state_dict`` key names — so reference checkpoints load unchanged — but every
``forward`` runs hand-written HIP kernels on an MI355X through libfnssl_hip.so.

    FNblock(input_size, hidden_size=256, dropout=0.2, is_online=False, is_first=False)
        .forward(x, nb_skip=None, fb_skip=None) -> (x, fb_skip, nb_skip)      Model.py:9,31,50
    FN_SSL(input_size=4, hidden_size=256, is_online=True, is_doa=False)
        .forward(x[nb', 4, nf, nt]) -> [nb', nt//12, 2*nf]                    Model.py:56,72,90
    FN_lightning().arch = FN_SSL()                                            Model.py:92-99

The ``nn.LSTM`` / ``nn.Linear`` sub-modules are kept purely as parameter
containers (names, shapes, ``load_state_dict``, ``.cuda()``); they are never
called.  Inputs must be ROCm tensors — there is no CPU fallback.

``eval()`` mode (the reference's dropout is the identity there) takes the
forward-only inference kernels and returns tensors without a ``grad_fn``.
``train()`` mode takes the reserve-saving kernels behind a
``torch.autograd.Function`` (``fnssl/autograd.py``): ``loss.backward()`` runs the
HIP BPTT / weight-gradient kernels and fills the parameters' ``.grad``, so the
reference's training loops (Lightning automatic optimisation + DDP,
main.py:149-157,286-288; ``Learner.train_epoch``, Learner.py:104-115) and any
``torch.optim`` optimizer work unchanged.  Dropout masks: see ``fnssl/autograd.py``.
"""
import torch
import torch.nn as nn

from fnssl import ops


def _lstm_streams(lstm: nn.LSTM, c0: int, c2: int, device, bf16: bool = False, pad_to: int = 0, wide: bool = False):
    """Pack an nn.LSTM's parameters into per-direction device weight streams.  ``bf16``: the bf16-MFMA
    stream (``wide``: the tile-ordered stream of the 32-sequences-per-wave kernels); ``pad_to`` zero-pads the input
    columns (the last c2 ones are the skip segment) to that width."""
    out = []
    for sfx in [""] + (["_reverse"] if lstm.bidirectional else []):
        w_ih = getattr(lstm, "weight_ih_l0" + sfx).detach().float()
        if pad_to and w_ih.shape[1] < pad_to:
            w_ih = torch.cat((w_ih, w_ih.new_zeros((w_ih.shape[0], pad_to - w_ih.shape[1]))), dim=1)
        args = (w_ih, getattr(lstm, "weight_hh_l0" + sfx).detach().float(),
                getattr(lstm, "bias_ih_l0" + sfx).detach().float(), getattr(lstm, "bias_hh_l0" + sfx).detach().float(),
                c0, c2, device)
        out.append((ops.pack_lstm_bf16w(*args) if wide else ops.pack_lstm_bf16(*args)) if bf16 else ops.pack_lstm(*args))
    return out


def _param_key(module: nn.Module):
    """Changes whenever a parameter is replaced, moved or modified in place."""
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in module.parameters())


def _is_bf16(module: nn.Module) -> bool:
    """True after ``module.bfloat16()``: bf16 parameters select the bf16-MFMA kernels (fp32 accumulate / tensors)."""
    return next(module.parameters()).dtype == torch.bfloat16


def _require_eval(m: nn.Module, what: str = "forward"):
    if m.training:
        raise RuntimeError("%s.%s: forward-only MI355X entry — call .eval() first (of the drop-in modules only "
                           "Model.FN_SSL / Model.FNblock carry an autograd graph in train mode)" % (type(m).__name__, what))


class FNblock(nn.Module):
    """Full-band BiLSTM over frequency followed by a narrow-band LSTM over time."""

    def __init__(self, input_size, hidden_size=256, dropout=0.2, is_online=False, is_first=False):
        super(FNblock, self).__init__()
        self.input_size = input_size
        self.full_hidden_size = hidden_size // 2
        self.is_first = is_first
        self.is_online = is_online
        if self.is_online:
            self.narr_hidden_size = hidden_size
        else:
            self.narr_hidden_size = hidden_size // 2
        self.dropout = dropout

        self.dropout_full = nn.Dropout(p=self.dropout)
        self.dropout_narr = nn.Dropout(p=self.dropout)
        self.fullLstm = nn.LSTM(input_size=self.input_size, hidden_size=self.full_hidden_size, batch_first=True,
                                bidirectional=True)
        if self.is_first:
            self.narrLstm = nn.LSTM(input_size=2 * self.full_hidden_size + self.input_size,
                                    hidden_size=self.narr_hidden_size, batch_first=True,
                                    bidirectional=not self.is_online)
        else:
            self.narrLstm = nn.LSTM(input_size=2 * self.full_hidden_size, hidden_size=self.narr_hidden_size,
                                    batch_first=True, bidirectional=not self.is_online)
        self._packed = None
        self._packed_key = None

    def _streams(self, device):
        key = (_param_key(self), str(device))
        if self._packed is None or self._packed_key != key:
            fh2 = 2 * self.full_hidden_size
            if _is_bf16(self):
                # bf16-MFMA streams; the 4 data channels of block 1 are zero-padded to one 16-channel block
                cp = (self.input_size + 15) // 16 * 16
                if self.is_first:
                    full = _lstm_streams(self.fullLstm, cp, 0, device, True, cp)
                    narr = _lstm_streams(self.narrLstm, fh2, cp, device, True, fh2 + cp)
                else:
                    full = _lstm_streams(self.fullLstm, self.input_size, 0, device, True)
                    narr = _lstm_streams(self.narrLstm, fh2, 0, device, True)
            else:
                full = _lstm_streams(self.fullLstm, self.input_size, 0, device)
                narr = _lstm_streams(self.narrLstm, fh2, self.input_size if self.is_first else 0, device)
            self._packed, self._packed_key = (full, narr), key
        return self._packed

    @ops.on_device
    def forward(self, x, nb_skip=None, fb_skip=None):
        """x [nb, nt, nf, C] -> (x [nb, nt, nf, Hn], fb_skip [nb*nt, nf, 2Hf], nb_skip [nb*nf, nt, Hn]).

        As in the reference the incoming ``nb_skip`` is ignored: it is recomputed
        from ``x`` (Model.py:34).  The returned ``x`` is a permuted view of the
        narrow-band output, exactly like the reference's (Model.py:49).

        ``train()`` mode: dropout_full / dropout_narr are active (Model.py:40,48) and the three outputs carry a
        ``grad_fn`` (``fnssl.autograd.FNblockTrainFunction``: BPTT + weight-gradient kernels on ``backward()``).
        """
        if self.training:
            from fnssl import autograd as _ag
            return _ag.fnblock_train_forward(self, x, fb_skip)
        if _is_bf16(self):
            raise RuntimeError("FNblock.forward: the bf16 path runs through FN_SSL.forward (blocks are fused there)")
        nb, nt, nf, nc = x.shape
        full_w, narr_w = self._streams(x.device)
        fh2 = 2 * self.full_hidden_size
        nh = self.narr_hidden_size * (1 if self.is_online else 2)
        if not self.is_first:
            if fb_skip is None:
                raise RuntimeError("FNblock: fb_skip is required unless is_first")
            fb_prev = fb_skip.reshape(nb, nt, nf, -1)
        from fnssl import train as _train                       # combine(): the fused element-wise sum kernel
        f = torch.empty((nb, nt, nf, fh2), dtype=torch.float32, device=x.device)
        # The residual sums of blocks 2 / 3 (Model.py:36-37, 44-45) are formed by one element-wise kernel each, so that every
        # LSTM call sees ONE summed input and takes the shape-specialised / cluster-resident kernels (a second summed operand
        # is only built into the generic rounds and the several-waves-per-group kernels); same fp32 add, same bits.
        if self.is_first:
            ops.lstm_layer("full", x, None, None, full_w, self.full_hidden_size, f)
        else:
            x = ops._conform(x.float())
            fb_prev = ops._conform(fb_prev.float())
            u = torch.empty((nb, nt, nf, nc), dtype=torch.float32, device=x.device)
            _train.combine(u, plain=(x, fb_prev))                                              # x + fb_skip  :36-37
            ops.lstm_layer("full", u, None, None, full_w, self.full_hidden_size, f)
        n = torch.empty((nb, nf, nt, nh), dtype=torch.float32, device=x.device)
        n_logical = n.permute(0, 2, 1, 3)                      # [nb, nt, nf, Hn] view
        if self.is_first:
            ops.lstm_layer("narrow", f, None, x, narr_w, self.narr_hidden_size, n_logical)     # cat  :42-43
        else:
            v = torch.empty((nb, nf, nt, fh2), dtype=torch.float32, device=x.device).permute(0, 2, 1, 3)
            _train.combine(v, plain=(f, x))                                                    # + nb_skip  :44-45
            ops.lstm_layer("narrow", v, None, None, narr_w, self.narr_hidden_size, n_logical)
        return n_logical, f.view(nb * nt, nf, fh2), n.view(nb * nf, nt, nh)


class FN_SSL(nn.Module):
    """Three FN blocks + AvgPool(12) . Linear(256, 2) . tanh head -> DP-IPD [cos | sin]."""

    def __init__(self, input_size=4, hidden_size=256, is_online=True, is_doa=False):
        super(FN_SSL, self).__init__()
        self.is_online = is_online
        self.is_doa = is_doa
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.block_1 = FNblock(input_size=self.input_size, is_online=self.is_online, is_first=True)
        self.block_2 = FNblock(input_size=self.hidden_size, is_online=self.is_online, is_first=False)
        self.block_3 = FNblock(input_size=self.hidden_size, is_online=self.is_online, is_first=False)
        self.emb2ipd = nn.Linear(256, 2)
        self.pooling = nn.AvgPool2d(kernel_size=(12, 1))
        self.tanh = nn.Tanh()
        if self.is_doa:
            self.ipd2doa = nn.Linear(512, 180)
        self._net = None
        self._net_key = None
        self.chunk_pairs = 0      # pairs per pass inside the library (0 = whole batch)
        # train mode (fnssl/autograd.py): dropout-mask keys.  seed None = torch.initial_seed(); pair_offset None =
        # rank * pairs under an initialised process group (DDP's equal shards), else 0
        self.dropout_seed, self.dropout_calls, self.pair_offset = None, 0, None
        for k, blk in enumerate((self.block_1, self.block_2, self.block_3)):
            blk.dropout_layer = 2 * k          # mask ids when a block is driven on its own

    def device_net(self, device) -> "ops.DeviceNet":
        key = (_param_key(self), str(device))
        if self._net is None or self._net_key != key:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._net = ops.DeviceNet(sd, device, self.is_online, self.is_doa, self.input_size)
            self._net_key = key
        return self._net

    @ops.on_device
    def forward_seq(self, x0):
        """x0 [nb', nt, nf, input_size] (the layout the front-end kernels emit) -> DP-IPD."""
        if self.training:
            from fnssl import autograd as _ag
            return _ag.fnssl_train_forward(self, x0.permute(0, 3, 2, 1).float())
        if _is_bf16(self):
            return self._forward_seq_bf16(x0.float()).to(x0.dtype)
        return self.device_net(x0.device).forward(x0, self.chunk_pairs)

    def _forward_seq_bf16(self, x0):
        """Fast mode after ``.bfloat16()``: the six LSTM products run on bf16 MFMAs (weights and [x | h] operands
        rounded to bf16, fp32 accumulation); residual adds, head and every tensor in HBM stay fp32.  NOT the
        reference's arithmetic — see DESIGN.md §9 for the measured deviation."""
        from fnssl import train as _train                       # combine(): fused element-wise sums
        nbp = x0.shape[0]
        cp = self.chunk_pairs if self.chunk_pairs > 0 else nbp
        outs = []
        for b0 in range(0, nbp, cp):
            x = x0[b0:b0 + cp]
            nb, nt, nf, c = x.shape
            dev = x.device
            xs = torch.cat((x, x.new_zeros((nb, nt, nf, (c + 15) // 16 * 16 - c))), dim=-1) if c % 16 else x
            cur, fb_prev = None, None
            for blk in (self.block_1, self.block_2, self.block_3):
                full_w, narr_w = blk._streams(dev)
                f = torch.empty((nb, nt, nf, 2 * blk.full_hidden_size), dtype=torch.float32, device=dev)
                nh = blk.narr_hidden_size * (1 if blk.is_online else 2)
                n = torch.empty((nb, nf, nt, nh), dtype=torch.float32, device=dev).permute(0, 2, 1, 3)
                if blk.is_first:
                    ops.lstm_layer("full", xs, None, None, full_w, blk.full_hidden_size, f, bf16=True)
                    ops.lstm_layer("narrow", f, None, xs, narr_w, blk.narr_hidden_size, n, bf16=True)     # cat :42-43
                else:
                    u = torch.empty_like(f)
                    _train.combine(u, plain=(cur, fb_prev))                                       # x + fb_skip :36-37
                    ops.lstm_layer("full", u, None, None, full_w, blk.full_hidden_size, f, bf16=True)
                    v = torch.empty((nb, nf, nt, f.shape[3]), dtype=torch.float32, device=dev).permute(0, 2, 1, 3)
                    _train.combine(v, plain=(f, cur))                                             # + nb_skip :44-45
                    ops.lstm_layer("narrow", v, None, None, narr_w, blk.narr_hidden_size, n, bf16=True)
                cur, fb_prev = n, f
            y = ops.head(cur.permute(0, 2, 1, 3), self.emb2ipd.weight.detach().float(), self.emb2ipd.bias.detach().float())
            if self.is_doa:
                y = ops.linear(y.reshape(-1, y.shape[-1]), self.ipd2doa.weight.detach().float().t().contiguous(),
                               self.ipd2doa.bias.detach().float()).reshape(nb, y.shape[1], -1)
            outs.append(y)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    @ops.on_device
    def forward(self, x):
        """x [nb', input_size, nf, nt] -> [nb', nt//12, 2*nf]  (or [.., 180] with is_doa).

        ``train()`` mode returns a tensor with a ``grad_fn`` (``fnssl.autograd.FNSSLTrainFunction``): the reference's
        ``pred = self(in_batch); loss.backward()`` (main.py:153-154, Learner.py:104-112) runs on the HIP kernels."""
        if x.ndim != 4 or x.shape[1] != self.input_size:
            raise RuntimeError("FN_SSL: expected [nb, %d, nf, nt], got %s" % (self.input_size, tuple(x.shape)))
        if self.training:
            from fnssl import autograd as _ag
            return _ag.fnssl_train_forward(self, x.float()).to(x.dtype)
        return self.forward_seq(ops.nchw_to_seq(x.float())).to(x.dtype)   # permute(0,3,2,1), Model.py:73


    @ops.on_device
    def forward_stream(self, x, state=None):
        """Streaming inference of the online (causal) model — the state carry the reference's causality
        permits but does not implement.

        x [nb', input_size, nf, T]: the NEXT T frames (T a positive multiple of 12); ``state`` is None for
        the first chunk, afterwards what the previous call returned.  Returns (DP-IPD [nb', T//12, 2*nf],
        state).  Consecutive chunks give bit-for-bit what ``forward`` gives on the whole signal: the
        full-band BiLSTMs only look along frequency inside a frame, the narrow-band LSTMs continue from
        their carried (h, c).
        """
        _require_eval(self, "forward_stream")
        if _is_bf16(self):
            raise RuntimeError("FN_SSL.forward_stream: fp32 model only (the streaming kernels are the exact-fp32 ones)")
        if not self.is_online:
            raise RuntimeError("FN_SSL.forward_stream: only the online model (uni-directional narrow-band) streams")
        if x.ndim != 4 or x.shape[1] != self.input_size or x.shape[3] == 0 or x.shape[3] % ops.SEG_FRAMES:
            raise RuntimeError("FN_SSL.forward_stream: expected [nb, %d, nf, T] with T a positive multiple of 12, got %s"
                               % (self.input_size, tuple(x.shape)))
        nb, _, nf, T = x.shape
        dev = x.device
        blocks = (self.block_1, self.block_2, self.block_3)
        if state is None:
            state = {"ws": [ops.lstm_state_workspace(nb * nf, b.narr_hidden_size, dev) for b in blocks],
                     "h": [None, None, None], "shape": (nb, nf), "frames": 0}
        elif state["shape"] != (nb, nf):
            raise RuntimeError("FN_SSL.forward_stream: batch / bins changed between chunks")
        from fnssl import train as _train                           # combine(): fused element-wise sums
        cur = ops.nchw_to_seq(x)                                    # [nb, T, nf, C]
        fb_prev = None
        for k, blk in enumerate(blocks):
            full_w, narr_w = blk._streams(dev)
            f = torch.empty((nb, T, nf, 2 * blk.full_hidden_size), dtype=torch.float32, device=dev)
            # The residual sums (Model.py:36-37, 44-45) are formed by one element-wise kernel each (a 12-frame chunk is a few
            # MB) so that the LSTM calls see ONE evenly spaced input tensor and take the slice-resident cluster kernels: a
            # chunk is 5 - 10 sequence groups, which the several-waves-per-group kernels serve by streaming the whole weight
            # matrix from L2 per group and step (29 -> ~8 ms per chunk, tools/latency_bench.py).  Same fp32 add either way.
            if blk.is_first:
                ops.lstm_layer("full", cur, None, None, full_w, blk.full_hidden_size, f)
            else:
                u = torch.empty((nb, T, nf, f.shape[3]), dtype=torch.float32, device=dev)
                _train.combine(u, plain=(cur, fb_prev))                                     # x + fb_skip  :36-37
                ops.lstm_layer("full", u, None, None, full_w, blk.full_hidden_size, f)
            nbuf = torch.empty((nb, nf, T + 1, blk.narr_hidden_size), dtype=torch.float32, device=dev)
            started = state["h"][k] is not None
            if started:
                nbuf[:, :, 0].copy_(state["h"][k])                  # h_{-1}: the row before the chunk's output
            out = nbuf[:, :, 1:].permute(0, 2, 1, 3)                # logical [nb, T, nf, Hn]
            if blk.is_first:
                ops.lstm_layer("narrow", f, None, cur, narr_w, blk.narr_hidden_size, out,
                               carry_workspace=state["ws"][k], carry=started)                    # cat  :42-43
            else:
                v = torch.empty((nb, nf, T, f.shape[3]), dtype=torch.float32, device=dev).permute(0, 2, 1, 3)
                _train.combine(v, plain=(f, cur))                                                 # + nb_skip  :44-45
                ops.lstm_layer("narrow", v, None, None, narr_w, blk.narr_hidden_size, out,
                               carry_workspace=state["ws"][k], carry=started)
            state["h"][k] = nbuf[:, :, T].clone()
            cur, fb_prev = out, f
        y = ops.head(nbuf[:, :, 1:].contiguous(), self.emb2ipd.weight.detach(), self.emb2ipd.bias.detach())
        if self.is_doa:
            y = ops.linear(y.reshape(-1, y.shape[-1]), self.ipd2doa.weight.detach().t().contiguous(),
                           self.ipd2doa.bias.detach()).reshape(nb, T // ops.SEG_FRAMES, -1)
        state["frames"] += T
        return y, state


class FN_lightning(nn.Module):
    """Wrapper whose ``arch.`` prefix matches Lightning checkpoints (Model.py:92-99)."""

    def __init__(self):
        super(FN_lightning, self).__init__()
        self.arch = FN_SSL()

    def forward(self, x):
        return self.arch(x)


if __name__ == "__main__":
    inp = torch.randn((2, 4, 256, 298)).cuda()
    net = FN_SSL().cuda().eval()
    out = net(inp)
    print(out.shape)
    print('# parameters:', sum(param.numel() for param in net.parameters()))
