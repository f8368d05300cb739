"""Drop-in for the reference's ``IPDnet/FixedAarryIPDnet.py`` (fixed-array IPDnet): same classes,
constructor arguments, ``forward`` signatures, return shapes and ``state_dict`` key names, with every
``forward`` running the HIP kernels of libfnssl_hip.so on an MI355X.

    FNblock(input_size, hidden_size=128, dropout=0.2, add_skip_dim=4, is_online=False, is_first=False)
        .forward(x, fb_skip, nb_skip) -> x                              FixedAarryIPDnet.py:11,29-40
    CausCnnBlock(inp_dim, out_dim, cnn_hidden_dim=128, ...)
        .forward(x[nb, C, nf, nt]) -> [nb, out_dim, nf, nt//12]         FixedAarryIPDnet.py:47,61-73
    IPDnet(input_size=4, hidden_size=128, max_track=2, is_online=True, n_seg=312)
        .forward(x[nb, 2*nch, nf, nt], offline_inference=False)
            -> [nb, nt//12, 2*nf, nch-1, max_track]                     FixedAarryIPDnet.py:80,91-120

Against the reference's dataflow nothing is concatenated or permuted in memory: the concat skips
(:34, :38) are a second operand segment of the consuming LSTM / conv kernel, the permutes are strides.
``nn.LSTM`` / ``nn.Conv2d`` sub-modules only hold parameters.  Forward-only (``eval()``), ROCm tensors
only.  fp32 by default; after ``net.bfloat16()`` (BASELINE config 3: bf16 weights, fp32 accumulate) the LSTM
layers run on bf16 MFMAs — weights and the [x | h] operands rounded to bf16, gates / cell state / every
tensor in HBM fp32 — and the conv head uses the bf16-rounded weights; inputs of either dtype are accepted and
the output comes back in the input's dtype.
"""
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from fnssl import ops                                               # noqa: E402
from Model import _lstm_streams, _param_key, _require_eval          # noqa: E402


def _split16(c):
    """Channels of a skip/input tensor as (vector channels, remainder channels) of the LSTM kernel."""
    return (c, 0) if c % 16 == 0 else (0, c)


def _ceil16(c):
    return (c + 15) // 16 * 16


def _aligned16(x):
    """Base address and batch / bin / time strides on 16-byte boundaries (what the LDS-DMA of conv_bf16x needs)."""
    if x is None:
        return True
    es = x.element_size()
    return x.data_ptr() % 16 == 0 and all((x.stride(d) * es) % 16 == 0 for d in range(3))


def _is_bf16(module):
    return next(module.parameters()).dtype == torch.bfloat16


def _pad_channels(t, c):
    """Zero-pad the last dimension to c channels (bf16 path: whole 16-channel blocks)."""
    if t.shape[-1] >= c:
        return t
    return torch.cat((t, t.new_zeros(tuple(t.shape[:-1]) + (c - t.shape[-1],))), dim=-1)


class FNblock(nn.Module):
    """Full-band BiLSTM over frequency and narrow-band LSTM over time with concatenated input skips."""

    def __init__(self, input_size, hidden_size=128, dropout=0.2, add_skip_dim=4, is_online=False, is_first=False):
        super(FNblock, self).__init__()
        self.input_size = input_size
        self.full_hidden_size = hidden_size // 2
        self.is_first = is_first
        self.is_online = is_online
        self.add_skip_dim = add_skip_dim
        if self.is_online:
            self.narr_hidden_size = hidden_size
        else:
            self.narr_hidden_size = hidden_size // 2
        self.dropout = dropout
        self.dropout_full = nn.Dropout(p=self.dropout)
        self.dropout_narr = nn.Dropout(p=self.dropout)
        if is_first:
            self.fullLstm = nn.LSTM(input_size=self.input_size, hidden_size=self.full_hidden_size,
                                    batch_first=True, bidirectional=True)
        else:
            self.fullLstm = nn.LSTM(input_size=self.input_size + add_skip_dim, hidden_size=self.full_hidden_size,
                                    batch_first=True, bidirectional=True)
        self.narrLstm = nn.LSTM(input_size=2 * self.full_hidden_size + add_skip_dim,
                                hidden_size=self.narr_hidden_size, batch_first=True,
                                bidirectional=not self.is_online)
        self._packed = None
        self._packed_key = None

    def _wide(self):
        """bf16 model at the hidden-256 (more than two microphones) layer shapes the 32-sequences-per-wave kernels
        (csrc/lstm_bf16w.h) are built for; FNSSL_NO_BF16W=1 keeps the 16-sequence kernels (A/B)."""
        return (_is_bf16(self) and not os.environ.get("FNSSL_NO_BF16W") and self.full_hidden_size == 128 and
                self.narr_hidden_size == 256 and self.is_online and _ceil16(self.add_skip_dim) == 16 and
                (self.input_size == 256 or (self.is_first and _ceil16(self.input_size) == 16)))

    def _streams(self, device):
        key = (_param_key(self), str(device), self._wide())
        if self._packed is None or self._packed_key != key:
            fh2 = 2 * self.full_hidden_size
            if self._wide():
                if self.is_first:
                    full = _lstm_streams(self.fullLstm, 16, 0, device, True, 16, wide=True)
                else:
                    full = _lstm_streams(self.fullLstm, self.input_size, 16, device, True, self.input_size + 16, wide=True)
                narr = _lstm_streams(self.narrLstm, fh2, 16, device, True, fh2 + 16, wide=True)
            elif _is_bf16(self):
                # the skip / network-input segment is zero-padded to whole 16-channel blocks
                cs = _ceil16(self.add_skip_dim)
                if self.is_first:
                    full = _lstm_streams(self.fullLstm, _ceil16(self.input_size), 0, device, True, _ceil16(self.input_size))
                else:
                    full = _lstm_streams(self.fullLstm, self.input_size, cs, device, True, self.input_size + cs)
                narr = _lstm_streams(self.narrLstm, fh2, cs, device, True, fh2 + cs)
            else:
                if self.is_first:
                    c0, c2 = _split16(self.input_size)
                else:
                    c0, c2 = self.input_size, self.add_skip_dim
                full = _lstm_streams(self.fullLstm, c0, c2, device)
                narr = _lstm_streams(self.narrLstm, fh2, self.add_skip_dim, device)
            self._packed, self._packed_key = (full, narr), key
        return self._packed

    @ops.on_device
    def run(self, x_main, x_skip, x_in=None):
        """Full-band input = [x_main | x_in] (first block: x_in alone), narrow-band input = [full out | x_skip];
        x_in defaults to x_skip (IPDnet.forward feeds the network input to both).  All operands are logical
        [nb, nt, nf, C] tensors with any strides.  Returns the narrow-band output, logical
        [nb, nt, nf, Hn] stored [nb, nf, nt, Hn]; its concatenation with the skip is left to the consumer."""
        _require_eval(self)
        if x_in is None:
            x_in = x_skip
        bf = _is_bf16(self)
        if bf:   # whole 16-channel blocks (no-ops when the caller already padded)
            x_skip = _pad_channels(x_skip, _ceil16(self.add_skip_dim))
            x_in = _pad_channels(x_in, _ceil16(x_in.shape[-1]))
        nb, nt, nf, _ = x_skip.shape
        full_w, narr_w = self._streams(x_skip.device)
        if self._wide():
            # activations travel as bf16 between the layers (what the next MFMA would round them to anyway)
            if x_main is not None and x_main.dtype != torch.bfloat16:
                x_main = x_main.to(torch.bfloat16)        # API-compat path (FNblock.forward) hands in fp32
            f = torch.empty((nb, nt, nf, 2 * self.full_hidden_size), dtype=torch.bfloat16, device=x_skip.device)
            if self.is_first:
                ops.lstm_layer("full", x_in, None, None, full_w, self.full_hidden_size, f, bf16=True, wide=True)
            else:
                ops.lstm_layer("full", x_main, None, x_in, full_w, self.full_hidden_size, f, bf16=True, wide=True)
            n = torch.empty((nb, nf, nt, self.narr_hidden_size), dtype=torch.bfloat16, device=x_skip.device)
            ops.lstm_layer("narrow", f, None, x_skip, narr_w, self.narr_hidden_size, n.permute(0, 2, 1, 3), bf16=True,
                           wide=True)
            return n.permute(0, 2, 1, 3)
        f = torch.empty((nb, nt, nf, 2 * self.full_hidden_size), dtype=torch.float32, device=x_skip.device)
        if self.is_first:
            if bf or _split16(self.input_size)[0]:
                ops.lstm_layer("full", x_in, None, None, full_w, self.full_hidden_size, f, bf16=bf)
            else:
                ops.lstm_layer("full", None, None, x_in, full_w, self.full_hidden_size, f)
        else:
            ops.lstm_layer("full", x_main, None, x_in, full_w, self.full_hidden_size, f, bf16=bf)   # cat :38 of the
        nh = self.narr_hidden_size * (1 if self.is_online else 2)                                    # block before
        n = torch.empty((nb, nf, nt, nh), dtype=torch.float32, device=x_skip.device)
        ops.lstm_layer("narrow", f, None, x_skip, narr_w, self.narr_hidden_size, n.permute(0, 2, 1, 3), bf16=bf)   # :34
        return n.permute(0, 2, 1, 3)

    @ops.on_device
    def forward(self, x, fb_skip, nb_skip):
        """Reference signature: x [nb, nt, nf, C], fb_skip [nb*nt, nf, Cs], nb_skip [nb*nf, nt, Cs]
        -> [nb, nt, nf, Hn + Cs] (the concatenation is materialised only here, for API compatibility)."""
        in_dtype = x.dtype
        x, fb_skip, nb_skip = x.float(), fb_skip.float(), nb_skip.float()
        nb, nt, nf, nc = x.shape
        skip = fb_skip.reshape(nb, nt, nf, -1)
        if self.is_first:
            n = self.run(None, skip, x)
        else:
            if nc != self.input_size + self.add_skip_dim:
                raise RuntimeError("FNblock: %d input channels, expected %d" % (nc, self.input_size + self.add_skip_dim))
            n = self.run(x[..., :self.input_size], skip, x[..., self.input_size:])
        nbs = nb_skip.reshape(nb, nf, nt, -1).permute(0, 2, 1, 3)
        return torch.cat((n.float(), nbs), dim=-1).to(in_dtype)


class CausCnnBlock(nn.Module):
    """conv3x3 -> ReLU -> pool 3 -> conv3x3 -> ReLU -> pool 4 -> conv3x3 -> tanh, causal in time."""

    def __init__(self, inp_dim, out_dim, cnn_hidden_dim=128, kernel=(3, 3), stride=(1, 1), padding=(1, 2)):
        super(CausCnnBlock, self).__init__()
        if tuple(kernel) != (3, 3) or tuple(stride) != (1, 1) or tuple(padding) != (1, 2):
            raise ValueError("CausCnnBlock: the MI355X path is built for kernel (3,3), stride (1,1), padding (1,2)")
        self.inp_dim, self.out_dim, self.cnn_hidden_dim = inp_dim, out_dim, cnn_hidden_dim
        self.conv1 = nn.Conv2d(inp_dim, cnn_hidden_dim, kernel_size=kernel, stride=stride, padding=padding, bias=False)
        self.conv2 = nn.Conv2d(cnn_hidden_dim, cnn_hidden_dim, kernel_size=kernel, stride=stride, padding=padding,
                               bias=False)
        self.conv3 = nn.Conv2d(cnn_hidden_dim, out_dim, kernel_size=kernel, stride=stride, padding=padding, bias=False)
        self.pooling1 = nn.AvgPool2d(kernel_size=(1, 3))
        self.pooling2 = nn.AvgPool2d(kernel_size=(1, 4))
        self.pad = padding
        self.relu = nn.ReLU(inplace=True)
        self.tanh = nn.Tanh()
        self._packed_x = None
        self._packed = None
        self._packed_key = None

    def _streams(self, device, ca, cb):
        key = (_param_key(self), str(device), ca, cb)
        if self._packed is None or self._packed_key != key:
            h = self.cnn_hidden_dim
            if h % 16:
                raise RuntimeError("CausCnnBlock: cnn_hidden_dim must be a multiple of 16")
            bf = _is_bf16(self)
            w1 = self.conv1.weight.detach().float()
            if w1.shape[1] < ca + cb:        # bf16 path: the skip segment arrives zero-padded to 16 channels
                w1 = torch.cat((w1, w1.new_zeros((w1.shape[0], ca + cb - w1.shape[1], 3, 3))), dim=1)
            self._packed = (ops.pack_conv3x3(w1, ca, cb, device, bf),
                            ops.pack_conv3x3(self.conv2.weight, h, 0, device, bf),
                            ops.pack_conv3x3(self.conv3.weight, h, 0, device, bf))
            # the LDS-staged kernel (conv_bf16x.hip) for the two large convolutions, where it takes the shapes
            self._packed_x = None
            if bf and ops.conv3x3_bf16x_supported(h, ca, cb) and ops.conv3x3_bf16x_supported(h, h, 0) \
                    and not os.environ.get("FNSSL_NO_CONVX"):
                self._packed_x = (ops.pack_conv3x3_bf16x(w1, ca, cb, device),
                                  ops.pack_conv3x3_bf16x(self.conv2.weight, h, 0, device))
            self._packed_key = key
        return self._packed

    @ops.on_device
    def run(self, xa, xb):
        """Channel concatenation [xa | xb] of logical [nb, nf, nt, C] tensors -> [nb, nf, nt//12, ceil4(out_dim)]."""
        _require_eval(self)
        bf = _is_bf16(self)
        if bf and xb is not None:
            xb = _pad_channels(xb, _ceil16(xb.shape[3]))          # whole 16-channel blocks (zero weights there)
        ca = xa.shape[3]
        cb = 0 if xb is None else xb.shape[3]
        if ca + cb != self.inp_dim and not (bf and ca + cb == _ceil16(self.inp_dim - ca) + ca):
            raise RuntimeError("CausCnnBlock: %d + %d input channels, expected %d" % (ca, cb, self.inp_dim))
        w1, w2, w3 = self._streams(xa.device, ca, cb)
        if self._packed_x is not None and xa.dtype == torch.bfloat16 and _aligned16(xa) and _aligned16(xb):
            x1, x2 = self._packed_x
            # conv -> ReLU -> AvgPool((1, 3)) and conv -> ReLU -> AvgPool((1, 4)), pooling in the conv epilogues
            y = ops.conv3x3_causal_bf16x(xa, xb, x1, self.cnn_hidden_dim, "relu", pool=3, bf16_out=True)
            y = ops.conv3x3_causal_bf16x(y, None, x2, self.cnn_hidden_dim, "relu", pool=4)
            return ops.conv3x3_causal(y, None, w3, self.out_dim, "tanh", bf)
        y = ops.conv3x3_causal(xa, xb, w1, self.cnn_hidden_dim, "relu", bf)
        y = ops.avgpool_time(y, 3)
        y = ops.conv3x3_causal(y, None, w2, self.cnn_hidden_dim, "relu", bf)
        y = ops.avgpool_time(y, 4)
        return ops.conv3x3_causal(y, None, w3, self.out_dim, "tanh", bf)

    @ops.on_device
    def forward(self, x):
        """Reference signature: x [nb, inp_dim, nf, nt] -> [nb, out_dim, nf, nt // 12]."""
        in_dtype = x.dtype
        c = x.shape[1]
        ca = c - c % 16
        xl = x.float().permute(0, 2, 3, 1).contiguous()              # channels-last (plumbing)
        if ca == 0:
            raise RuntimeError("CausCnnBlock: needs at least 16 input channels")
        xa = xl[..., :ca]
        xb = xl[..., ca:] if ca < c else None
        y = self.run(xa, xb)
        return y[..., :self.out_dim].permute(0, 3, 1, 2).to(in_dtype)


class IPDnet(nn.Module):
    """Fixed-array IPDnet: two FN blocks with input concat-skips and the causal conv head."""

    def __init__(self, input_size=4, hidden_size=128, max_track=2, is_online=True, n_seg=312):
        super(IPDnet, self).__init__()
        self.is_online = is_online
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.block_1 = FNblock(input_size=self.input_size, hidden_size=self.hidden_size,
                               add_skip_dim=self.input_size, is_online=self.is_online, is_first=True)
        self.block_2 = FNblock(input_size=self.hidden_size, hidden_size=self.hidden_size,
                               add_skip_dim=self.input_size, is_online=self.is_online, is_first=False)
        self.cnn_out_dim = 2 * ((input_size // 2) - 1) * max_track
        self.cnn_inp_dim = hidden_size + input_size
        self.conv = CausCnnBlock(inp_dim=self.cnn_inp_dim, out_dim=self.cnn_out_dim)
        self.n = n_seg
        self._side = None                                           # streams of the two half-batches (bf16 wide path)

    def _two_streams(self, nb):
        """Opt-in (``FNSSL_IPDNET_STREAMS=2``): half-batches on separate streams.  It was the default while the pair-split
        LSTM kernels left partial rounds (600 full-band workgroups on 256 CUs); with the cluster-resident kernels every
        launch fills the chip by itself and one stream is both faster (24.9-25.0 against 25.3-25.4 ms at config 3) and
        deterministic in its timing."""
        try:
            ns = int(os.environ.get("FNSSL_IPDNET_STREAMS", "1"))
        except ValueError:
            ns = 1
        return (_is_bf16(self) and self.block_1._wide() and nb >= 8 and ns >= 2
                and not os.environ.get("FNSSL_IPDNET_ONE_STREAM"))

    @ops.on_device
    def forward(self, x, offline_inference=False):
        _require_eval(self)
        in_dtype = x.dtype
        x = x.float()                                               # tensors in HBM are fp32 in both precisions
        nb, nc, nf, nt = x.shape
        if nc != self.input_size:
            raise RuntimeError("IPDnet: %d input channels, expected %d" % (nc, self.input_size))
        ou_frame = nt // 12
        nseg = 1
        if not self.is_online and offline_inference:
            # chunk-wise offline inference (:96-100): zero-pad the time axis to a multiple of n and
            # fold the segments into the batch
            pad = (self.n - nt % self.n) % self.n
            if pad:
                x = torch.cat((x, x.new_zeros((nb, nc, nf, pad))), dim=3)
            nseg = (nt + pad) // self.n
            x = x.reshape(nb, nc, nf, nseg, self.n).permute(0, 3, 1, 2, 4).reshape(nb * nseg, nc, nf, self.n)
            nb, nt = nb * nseg, self.n
        xs = ops.nchw_to_seq(x)                                     # [nb, nt, nf, C]  (:93)
        xp = _pad_channels(xs, _ceil16(nc)) if _is_bf16(self) else xs   # bf16 kernels read whole 16-channel blocks

        def trunk(xq):
            y = self.block_1.run(None, xq)
            y = self.block_2.run(y, xq)                             # logical [nb, nt, nf, Hn], stored [nb, nf, nt, Hn]
            xc = xq if _is_bf16(self) else xq[..., :nc]           # (bf16: the padded channels meet zero weights)
            return self.conv.run(y.permute(0, 2, 1, 3), xc.permute(0, 2, 1, 3))   # [nb, nf, nt2, ceil4(Cout)]

        if self._two_streams(nb):
            # Utterances are independent: parts of the batch go down separate streams, so that one part's partial round
            # can share the chip with the other part's kernels.  Same kernels, same results (batch slices).
            cur = torch.cuda.current_stream()
            ns = max(2, min(int(os.environ.get("FNSSL_IPDNET_STREAMS", "2")), nb))
            if self._side is None or len(self._side) != ns or self._side[0].device != x.device:
                self._side = tuple(torch.cuda.Stream(device=x.device) for _ in range(ns))   # keyed by (device, count)
            # pack / upload the weight streams on the CALLER's stream before forking: the side streams only read them
            self.block_1._streams(x.device)
            self.block_2._streams(x.device)
            self.conv._streams(x.device, self.block_2.narr_hidden_size, xp.shape[3])   # two-stream path: online, bf16
            bounds = [nb * i // ns for i in range(ns + 1)]
            parts = []
            for st, lo, hi in zip(self._side, bounds[:-1], bounds[1:]):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    parts.append(trunk(xp[lo:hi]))
            for st, cpart in zip(self._side, parts):
                cur.wait_stream(st)
                cpart.record_stream(cur)
            c = torch.cat(parts, dim=0)
        else:
            c = trunk(xp)
        nt2 = nt // 12
        c = c[..., :self.cnn_out_dim].permute(0, 2, 1, 3)           # = conv(x).permute(0,3,2,1)  (:113)
        c = c.reshape(nb, nt2, nf, 2, -1).permute(0, 1, 3, 2, 4)
        if nseg > 1 or (not self.is_online and offline_inference):
            c = c.reshape(nb // nseg, nt2 * nseg, 2, nf * 2, -1).permute(0, 1, 3, 4, 2)
            return c[:, :ou_frame, :, :, :].to(in_dtype)
        return c.reshape(nb, nt2, 2, nf * 2, -1).permute(0, 1, 3, 4, 2).to(in_dtype)

    @ops.on_device
    def forward_stream(self, x, state=None):
        """Streaming inference of the online model (the state carry SURVEY.md §8f rank 3 asks for; the reference's
        causality permits it but it has no such entry): x [nb, 2*nch, nf, T] = the NEXT T frames (T a positive
        multiple of 12), ``state`` = None for the first chunk, then what the previous call returned.  Returns
        (IPD [nb, T//12, 2*nf, nch-1, max_track], state).  Consecutive chunks reproduce ``forward`` on the whole
        signal bit for bit: the narrow-band LSTMs continue from their carried (h, c), every causal conv sees the
        last two frames of its previous input, and the 3 / 4-frame poolings stay aligned because T % 12 == 0."""
        _require_eval(self)
        if not self.is_online or _is_bf16(self):
            raise RuntimeError("IPDnet.forward_stream: online fp32 model only")
        if x.ndim != 4 or x.shape[1] != self.input_size or x.shape[3] == 0 or x.shape[3] % 12:
            raise RuntimeError("IPDnet.forward_stream: expected [nb, %d, nf, T] with T a positive multiple of 12, got %s"
                               % (self.input_size, tuple(x.shape)))
        x = x.float()
        nb, nc, nf, T = x.shape
        dev = x.device
        blocks = (self.block_1, self.block_2)
        hid = self.conv.cnn_hidden_dim
        if state is None:
            state = {"ws": [ops.lstm_state_workspace(nb * nf, b.narr_hidden_size, dev) for b in blocks],
                     "n_tail": [torch.zeros((nb, nf, 2, b.narr_hidden_size), device=dev) for b in blocks],
                     "x_tail": torch.zeros((nb, 2, nf, nc), device=dev),
                     "p_tail": [torch.zeros((nb, nf, 2, hid), device=dev) for _ in range(2)],
                     "shape": (nb, nf), "frames": 0}
        elif state["shape"] != (nb, nf):
            raise RuntimeError("IPDnet.forward_stream: batch / bins changed between chunks")
        started = state["frames"] > 0
        xbuf = torch.empty((nb, T + 2, nf, nc), dtype=torch.float32, device=dev)
        xbuf[:, :2].copy_(state["x_tail"])
        xbuf[:, 2:].copy_(ops.nchw_to_seq(x))                      # [nb, T, nf, C] behind the 2 carried frames
        xs = xbuf[:, 2:]
        cur = None
        for k, blk in enumerate(blocks):
            full_w, narr_w = blk._streams(dev)
            f = torch.empty((nb, T, nf, 2 * blk.full_hidden_size), dtype=torch.float32, device=dev)
            if blk.is_first:
                if _split16(blk.input_size)[0]:
                    ops.lstm_layer("full", xs, None, None, full_w, blk.full_hidden_size, f)
                else:
                    ops.lstm_layer("full", None, None, xs, full_w, blk.full_hidden_size, f)
            else:
                ops.lstm_layer("full", cur, None, xs, full_w, blk.full_hidden_size, f)
            nbuf = torch.empty((nb, nf, T + 2, blk.narr_hidden_size), dtype=torch.float32, device=dev)
            nbuf[:, :, :2].copy_(state["n_tail"][k])               # the last two output frames of the previous chunk
            out = nbuf[:, :, 2:].permute(0, 2, 1, 3)
            ops.lstm_layer("narrow", f, None, xs, narr_w, blk.narr_hidden_size, out,
                           carry_workspace=state["ws"][k], carry=started)
            state["n_tail"][k] = nbuf[:, :, T:].clone()
            cur = out
        state["x_tail"] = xbuf[:, T:].clone()
        ca, cb = nbuf.shape[3], nc
        w1, w2, w3 = self.conv._streams(dev, ca, cb)
        y = ops.conv3x3_causal(nbuf, xbuf.permute(0, 2, 1, 3), w1, hid, "relu")[:, :, 2:]      # drop the carried frames
        p1 = ops.avgpool_time(y, 3)
        in2 = torch.cat((state["p_tail"][0], p1), dim=2)
        y = ops.conv3x3_causal(in2, None, w2, hid, "relu")[:, :, 2:]
        state["p_tail"][0] = in2[:, :, -2:].clone()
        p2 = ops.avgpool_time(y, 4)
        in3 = torch.cat((state["p_tail"][1], p2), dim=2)           # a 12-frame chunk adds ONE frame here
        c = ops.conv3x3_causal(in3, None, w3, self.cnn_out_dim, "tanh")[:, :, 2:]
        state["p_tail"][1] = in3[:, :, -2:].clone()
        state["frames"] += T
        nt2 = T // 12
        c = c[..., :self.cnn_out_dim].permute(0, 2, 1, 3)
        c = c.reshape(nb, nt2, nf, 2, -1).permute(0, 1, 3, 2, 4)
        return c.reshape(nb, nt2, 2, nf * 2, -1).permute(0, 1, 3, 4, 2), state

