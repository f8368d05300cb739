"""Deterministic parameter generator for the FN-SSL DP-IPD network.

There is no network access and no checkpoint blob, so benches, tests and golden
fixtures all draw weights from this generator.  Names and shapes are the
reference's ``state_dict`` keys (reference: FN-SSL/Model.py:25-29 for the two
LSTMs of a block, :67/:70-71 for ``emb2ipd`` / ``ipd2doa``); the distribution is
PyTorch's default init (uniform +-1/sqrt(H) for nn.LSTM, +-1/sqrt(fan_in) for
nn.Linear) so activations sit in the same range a freshly built reference
model produces.  Values come from numpy's legacy ``RandomState`` stream, which
is stable across numpy versions, so a seed pins the exact bits.
"""
from __future__ import annotations

import collections

import numpy as np

GATES = 4  # i, f, g, o  (PyTorch order)


def lstm_param_shapes(input_size: int, hidden: int, bidirectional: bool):
    """(suffix, shape) list for one nn.LSTM(input_size, hidden, 1 layer)."""
    out = []
    for sfx in ([""] + (["_reverse"] if bidirectional else [])):
        out += [
            ("weight_ih_l0" + sfx, (GATES * hidden, input_size)),
            ("weight_hh_l0" + sfx, (GATES * hidden, hidden)),
            ("bias_ih_l0" + sfx, (GATES * hidden,)),
            ("bias_hh_l0" + sfx, (GATES * hidden,)),
        ]
    return out


def fnblock_param_shapes(input_size: int, hidden_size: int = 256, is_online: bool = False,
                         is_first: bool = False):
    """Parameter (name, shape) list of one FNblock (reference FN-SSL/Model.py:9-29)."""
    full_h = hidden_size // 2
    narr_h = hidden_size if is_online else hidden_size // 2
    narr_in = 2 * full_h + (input_size if is_first else 0)
    out = [("fullLstm." + n, s) for n, s in lstm_param_shapes(input_size, full_h, True)]
    out += [("narrLstm." + n, s) for n, s in lstm_param_shapes(narr_in, narr_h, not is_online)]
    return out


def fnssl_param_shapes(input_size: int = 4, hidden_size: int = 256, is_online: bool = True,
                       is_doa: bool = False):
    """Parameter (name, shape) list of FN_SSL (reference FN-SSL/Model.py:56-71).

    Like the reference, the three blocks are always built with the FNblock
    default hidden_size=256; ``hidden_size`` only sets the input width of
    blocks 2 and 3 (so anything but 256 is unusable there, exactly as upstream).
    """
    out = []
    for idx, (isz, first) in enumerate(
            [(input_size, True), (hidden_size, False), (hidden_size, False)], start=1):
        out += [("block_%d.%s" % (idx, n), s)
                for n, s in fnblock_param_shapes(isz, 256, is_online, first)]
    out += [("emb2ipd.weight", (2, 256)), ("emb2ipd.bias", (2,))]
    if is_doa:
        out += [("ipd2doa.weight", (180, 512)), ("ipd2doa.bias", (180,))]
    return out


def _bound(name: str, shape) -> float:
    if "_l0" in name:          # nn.LSTM parameter
        hidden = shape[0] // GATES
        return 1.0 / np.sqrt(hidden)
    if name.endswith("weight"):
        return 1.0 / np.sqrt(shape[1])
    # Linear bias: fan_in of the matching weight
    return 1.0 / np.sqrt({"emb2ipd.bias": 256, "ipd2doa.bias": 512}[name])


def make_state(shapes, seed: int = 0, scale: float = 1.0):
    """Ordered dict name -> float32 ndarray drawn from RandomState(seed)."""
    rs = np.random.RandomState(seed)
    sd = collections.OrderedDict()
    for name, shape in shapes:
        b = _bound(name, shape) * scale
        sd[name] = rs.uniform(-b, b, size=shape).astype(np.float32)
    return sd


def make_fnssl_state(seed: int = 0, input_size: int = 4, hidden_size: int = 256,
                     is_online: bool = True, is_doa: bool = False, scale: float = 1.0):
    return make_state(fnssl_param_shapes(input_size, hidden_size, is_online, is_doa), seed, scale)


def make_fnblock_state(seed: int, input_size: int, hidden_size: int = 256,
                       is_online: bool = False, is_first: bool = False, scale: float = 1.0):
    return make_state(fnblock_param_shapes(input_size, hidden_size, is_online, is_first), seed, scale)


def n_params(sd) -> int:
    return int(sum(v.size for v in sd.values()))


# --------------------------------------------------------------------------- #
# IPDnet (fixed array), reference IPDnet/FixedAarryIPDnet.py:7-90
# --------------------------------------------------------------------------- #
def ipdnet_param_shapes(input_size: int = 4, hidden_size: int = 128, max_track: int = 2, is_online: bool = True):
    """(name, shape) list of IPDnet: two concat-skip FN blocks + a 3-layer causal Conv2d head."""
    fh = hidden_size // 2
    nh = hidden_size if is_online else hidden_size // 2
    out = []
    for bi, full_in in ((1, input_size), (2, hidden_size + input_size)):
        out += [("block_%d.fullLstm.%s" % (bi, n), s) for n, s in lstm_param_shapes(full_in, fh, True)]
        out += [("block_%d.narrLstm.%s" % (bi, n), s)
                for n, s in lstm_param_shapes(2 * fh + input_size, nh, not is_online)]
    cout = 2 * ((input_size // 2) - 1) * max_track
    cin = hidden_size + input_size
    out += [("conv.conv1.weight", (128, cin, 3, 3)), ("conv.conv2.weight", (128, 128, 3, 3)),
            ("conv.conv3.weight", (cout, 128, 3, 3))]
    return out


def make_ipdnet_state(seed: int = 0, input_size: int = 4, hidden_size: int = 128, max_track: int = 2,
                      is_online: bool = True):
    rs = np.random.RandomState(seed)
    sd = collections.OrderedDict()
    for name, shape in ipdnet_param_shapes(input_size, hidden_size, max_track, is_online):
        if "_l0" in name:
            b = 1.0 / np.sqrt(shape[0] // GATES)
        else:                                   # Conv2d default init: U(+-1/sqrt(fan_in))
            b = 1.0 / np.sqrt(shape[1] * shape[2] * shape[3])
        sd[name] = rs.uniform(-b, b, size=shape).astype(np.float32)
    return sd


# --------------------------------------------------------------------------- #
# IPDnet2 (OnlineSpatialNet), reference IPDnet2/IPDnet2.py:85-136 (layer), :259-330 (network)
# --------------------------------------------------------------------------- #
def mamba_param_shapes(d_model: int, d_state: int = 16, d_conv: int = 4, expand: int = 2):
    """(name, shape) list of one ``mamba_ssm.Mamba(d_model, d_state, d_conv)`` block (the names a real
    checkpoint of the reference carries; call sites IPDnet2/IPDnet2.py:127,132)."""
    d_inner = expand * d_model
    dt_rank = -(-d_model // 16)
    return [("in_proj.weight", (2 * d_inner, d_model)), ("conv1d.weight", (d_inner, 1, d_conv)),
            ("conv1d.bias", (d_inner,)), ("x_proj.weight", (dt_rank + 2 * d_state, d_inner)),
            ("dt_proj.weight", (d_inner, dt_rank)), ("dt_proj.bias", (d_inner,)),
            ("A_log", (d_inner, d_state)), ("D", (d_inner,)), ("out_proj.weight", (d_model, d_inner))]


def ipdnet2_param_shapes(dim_input: int = 10, dim_output: int = 16, num_layers: int = 8, dim_hidden: int = 96,
                         dim_squeeze: int = 8, num_freqs: int = 256, encoder_kernel_size: int = 5,
                         f_kernel_size: int = 5, f_groups: int = 8, d_state: int = 16, d_conv: int = 4,
                         fre_compression_ratio: int = 16):
    """(name, shape) list of OnlineSpatialNet with attention='mamba(d_state,d_conv)' and all-LN norms."""
    H = dim_hidden
    out = [("encoder.weight", (H, dim_input, encoder_kernel_size)), ("encoder.bias", (H,))]
    for l in range(num_layers):
        p = "layers.%d." % l
        nfull = num_freqs // 2 if l == 0 else num_freqs // fre_compression_ratio
        for fc in ("fconv1", "fconv2"):
            out += [(p + fc + ".0.weight", (H,)), (p + fc + ".0.bias", (H,)),
                    (p + fc + ".1.weight", (H, H // f_groups, f_kernel_size)), (p + fc + ".1.bias", (H,)),
                    (p + fc + ".2.weight", (H,))]
        out += [(p + "norm_full.weight", (H,)), (p + "norm_full.bias", (H,)),
                (p + "squeeze.0.weight", (dim_squeeze, H, 1)), (p + "squeeze.0.bias", (dim_squeeze,)),
                (p + "full.weight", (nfull, nfull)), (p + "full.bias", (nfull,)),
                (p + "unsqueeze.0.weight", (H, dim_squeeze, 1)), (p + "unsqueeze.0.bias", (H,))]
        for nm, mm in (("norm_mhsa", "mhsa"), ("norm_tconvffn", "tconvffn")):
            out += [(p + nm + ".weight", (H,)), (p + nm + ".bias", (H,))]
            out += [(p + mm + "." + n, s) for n, s in mamba_param_shapes(H, d_state, d_conv)]
    out += [("freq_inverse.trans2.weight", (fre_compression_ratio * dim_output, H, 1)),
            ("freq_inverse.trans2.bias", (fre_compression_ratio * dim_output,)),
            ("decoder.weight", (dim_output, dim_output)), ("decoder.bias", (dim_output,))]
    return out


def make_ipdnet2_state(seed: int = 0, **cfg):
    """Deterministic OnlineSpatialNet parameters: PyTorch-default-like ranges for conv / linear layers, norm
    gains around 1, PReLU slopes around 0.25, and Mamba's own init ranges (A = -exp(A_log) in [-16, -0.5],
    D around 1, softplus(dt bias) in [1e-3, 1e-1])."""
    rs = np.random.RandomState(seed)
    sd = collections.OrderedDict()
    for name, shape in ipdnet2_param_shapes(**cfg):
        leaf = name.split(".")[-1]
        if name.endswith("A_log"):
            v = np.log(rs.uniform(0.5, 16.0, size=shape))
        elif leaf == "D":
            v = rs.uniform(0.5, 1.5, size=shape)
        elif name.endswith("dt_proj.bias"):
            dt = np.exp(rs.uniform(np.log(1e-3), np.log(1e-1), size=shape))
            v = dt + np.log(-np.expm1(-dt))                    # inverse softplus
        elif name.endswith("dt_proj.weight"):
            b = shape[1] ** -0.5
            v = rs.uniform(-b, b, size=shape)
        elif ".fconv" in name and name.endswith(".2.weight"):   # PReLU slope
            v = rs.uniform(0.1, 0.4, size=shape)
        elif len(shape) == 1 and leaf == "weight":              # LayerNorm gain
            v = rs.uniform(0.7, 1.3, size=shape)
        elif len(shape) == 1:                                   # biases
            v = rs.uniform(-0.1, 0.1, size=shape)
        else:                                                   # conv / linear weight: U(+-1/sqrt(fan_in))
            b = 1.0 / np.sqrt(np.prod(shape[1:]))
            v = rs.uniform(-b, b, size=shape)
        sd[name] = np.asarray(v, dtype=np.float32)
    return sd
