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
