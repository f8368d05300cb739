"""ctypes binding of csrc/libfnssl_hip.so (declarations: include/fnssl.h).

The library must share the HIP runtime instance PyTorch uses (streams and
device pointers are passed across), so ``torch`` is imported first: its bundled
``libamdhip64.so`` has SONAME ``libamdhip64.so.7`` and the dynamic loader then
resolves the library's dependency to that already-loaded object.  ``load()``
verifies that exactly one HIP runtime is mapped.

There is no fallback: if the shared object is missing this raises, and every
op in ``fnssl.ops`` refuses non-ROCm tensors.

Tuning.  The library reads no environment variable; kernel-family overrides and
A/B knobs travel in a caller-owned ``fnssl_tuning`` (include/fnssl.h).  This
binding is where ``FNSSL_<KNOB>`` environment variables are parsed — ONCE, when
the library is loaded, and again whenever ``refresh_tuning()`` is called (the
A/B legs of bench.py and the tests call it after changing ``os.environ``) — and
handed to the library with ``fnssl_tuning_set`` (the process default).
``tuning(**knobs)`` is the explicit form: a context manager that sets knobs for
the calls made inside it.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libfnssl_hip.so")
# development only: FNSSL_LIB_PATH loads another build of the SAME library (e.g. csrc/libfnssl_hip_abl.so, the
# `make ABLATE=1` build with the timing-ablation twins) — never a different implementation, never a fallback
LIB_PATH = os.environ.get("FNSSL_LIB_PATH", LIB_PATH)

ABI_VERSION = 19
CH_MODE = {"M": 0, "MM": 1}
# kernel families fnssl_lstm_plan reports (include/fnssl.h: FNSSL_LSTM_FAMILY_*)
LSTM_FAMILY = {1: "generic", 2: "static", 3: "static2 (retired)", 4: "split", 5: "split_static", 6: "f32_cluster",
               7: "static3", 8: "bf16", 9: "bf16_solo", 10: "bf16_pair", 11: "bf16_cluster", 12: "train", 13: "bwd", 14: "bwd_cluster"}

# every symbol include/fnssl.h declares
SYMBOLS = [
    "fnssl_abi_version", "fnssl_last_error", "fnssl_tuning_set", "fnssl_tuning_get", "fnssl_tuning_name", "fnssl_occupy_cus", "fnssl_num_frames", "fnssl_num_pairs", "fnssl_stft",
    "fnssl_num_frames_ex", "fnssl_stft_ex", "fnssl_array_frontend",
    "fnssl_forgetting_coefs", "fnssl_pair_features", "fnssl_nchw_to_seq", "fnssl_lstm_packed_floats",
    "fnssl_lstm_pack", "fnssl_lstm_workspace_bytes", "fnssl_lstm_workspace_bytes_ex", "fnssl_lstm_plan_rounds", "fnssl_lstm_plan", "fnssl_lstm_forward", "fnssl_lstm_cluster_status", "fnssl_head", "fnssl_linear",
    "fnssl_ipd2doa", "fnssl_doa_peaks", "fnssl_dpipd_targets", "fnssl_conv3x3_packed_floats", "fnssl_conv3x3_pack", "fnssl_conv3x3_causal",
    "fnssl_avgpool_time", "fnssl_array_features", "fnssl_conv3x3_packed_floats_bf16", "fnssl_conv3x3_pack_bf16",
    "fnssl_conv3x3_causal_bf16", "fnssl_conv3x3_causal_bf16a",
    "fnssl_conv3x3_packed_bytes_bf16x", "fnssl_conv3x3_pack_bf16x", "fnssl_conv3x3_causal_bf16x",
    "fnssl_avgpool_time_bf16",
    "fnssl_lstm_reserve_bytes", "fnssl_lstm_bwd_packed_floats", "fnssl_lstm_pack_bwd", "fnssl_lstm_bwd_workspace_bytes",
    "fnssl_lstm_backward", "fnssl_lstm_backward_plan", "fnssl_lstm_backward_status", "fnssl_lstm_weight_grads_workspace_bytes", "fnssl_lstm_weight_grads", "fnssl_lstm_packed_floats_bf16", "fnssl_lstm_pack_bf16", "fnssl_train_combine", "fnssl_dropout_scale", "fnssl_head_backward_workspace_bytes",
    "fnssl_head_backward", "fnssl_mse_loss", "fnssl_adam_step",
    "fnssl_forward_workspace_bytes", "fnssl_forward", "fnssl_timing_enable", "fnssl_timing_collect", "fnssl_timing_select", "fnssl_mfma_f32_peak", "fnssl_mfma_f32_peak_clocks",
    "fnssl_lstm_packed_floats_bf16w", "fnssl_lstm_pack_bf16w",
    "fnssl_train_create", "fnssl_train_destroy", "fnssl_train_param_floats", "fnssl_train_param_offset",
    "fnssl_train_map_bytes", "fnssl_train_upload_maps", "fnssl_train_workspace_bytes", "fnssl_train_backward",
    "fnssl_train_step",
    "fnssl_sn_layernorm", "fnssl_sn_encoder", "fnssl_sn_fconv", "fnssl_sn_full", "fnssl_sn_mamba_workspace_bytes",
    "fnssl_sn_mamba", "fnssl_sn_head", "fnssl_sn_forward_workspace_bytes", "fnssl_sn_state_floats", "fnssl_sn_forward",
]


TUNE_COUNT = 48


class Tuning(C.Structure):
    """fnssl_tuning: knob[i] = 0 is every knob's default (names: ``tuning_names()``)."""
    _fields_ = [("struct_bytes", C.c_uint), ("knob", C.c_int * TUNE_COUNT)]


class View(C.Structure):
    _fields_ = [("p", C.c_void_p), ("so", C.c_longlong), ("si", C.c_longlong), ("st", C.c_longlong)]


class LstmDesc(C.Structure):
    _fields_ = [
        ("src0", View), ("src1", View), ("src2", View),
        ("c0", C.c_int), ("c2", C.c_int),
        ("out", C.c_void_p),
        ("out_so", C.c_longlong), ("out_si", C.c_longlong), ("out_st", C.c_longlong),
        ("skip", View), ("out_sum", C.c_void_p),
        ("hidden", C.c_int), ("ndir", C.c_int), ("nseq", C.c_int), ("q_inner", C.c_int), ("nsteps", C.c_int),
        ("wpack", C.c_void_p * 2),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("variant", C.c_int),
        ("reserve", C.c_void_p), ("reserve_bytes", C.c_size_t),
        ("carry_state", C.c_int),
        ("precision", C.c_int),
        ("f32_mask", C.c_int),
        ("fallback_count", C.c_void_p),
        ("tuning", C.POINTER(Tuning)),
    ]


class LstmBwdDesc(C.Structure):
    _fields_ = [
        ("reserve", C.c_void_p),
        ("dh", View),
        ("da", C.c_void_p), ("da_so", C.c_longlong), ("da_si", C.c_longlong), ("da_st", C.c_longlong),
        ("dx", C.c_void_p), ("dx_so", C.c_longlong), ("dx_si", C.c_longlong), ("dx_st", C.c_longlong),
        ("c0g", C.c_int),
        ("hidden", C.c_int), ("ndir", C.c_int), ("nseq", C.c_int), ("q_inner", C.c_int), ("nsteps", C.c_int),
        ("wpack_bwd", C.c_void_p * 2),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("fallback_count", C.c_void_p),
        ("tuning", C.POINTER(Tuning)),
    ]


class WgradDesc(C.Structure):
    _fields_ = [("da", C.c_void_p), ("lda", C.c_longlong), ("x0", C.c_void_p), ("ldx0", C.c_longlong), ("c0", C.c_int),
                ("x2", C.c_void_p), ("ldx2", C.c_longlong), ("c2", C.c_int), ("h", C.c_void_p), ("ldh", C.c_longlong),
                ("nseq", C.c_longlong), ("nsteps", C.c_int), ("hidden", C.c_int), ("ndir", C.c_int),
                ("g_wih", C.c_void_p * 2), ("g_whh", C.c_void_p * 2), ("g_bih", C.c_void_p * 2), ("g_bhh", C.c_void_p * 2),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class BtfView(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sb", C.c_longlong), ("st", C.c_longlong), ("sf", C.c_longlong)]


class Net(C.Structure):
    _fields_ = [
        ("wpack", ((C.c_void_p * 2) * 2) * 3),
        ("emb_w", C.c_void_p), ("emb_b", C.c_void_p),
        ("doa_wt", C.c_void_p), ("doa_b", C.c_void_p),
        ("input_size", C.c_int), ("is_online", C.c_int),
        ("fallback_count", C.c_void_p),
        ("tuning", C.POINTER(Tuning)),
    ]


SN_MAX_LAYERS = 16


class SnFconvW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_w", "ln_b", "wT", "bias", "prelu")]


class SnFullW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_w", "ln_b", "wsT", "bs", "wfT", "bf", "wuT", "bu")]


class SnMambaW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_w", "ln_b", "winT", "conv_w", "conv_b", "wxT", "wdt", "bdt", "a", "d",
                                          "woT")]


class SnLayer(C.Structure):
    _fields_ = [("fconv1", SnFconvW), ("fconv2", SnFconvW), ("full", SnFullW), ("mamba", SnMambaW * 2)]


class SnNet(C.Structure):
    _fields_ = [("dim_input", C.c_int), ("num_layers", C.c_int), ("time_ratio", C.c_int),
                ("enc_wT", C.c_void_p), ("enc_b", C.c_void_p),
                ("layers", SnLayer * SN_MAX_LAYERS),
                ("wfiP", C.c_void_p), ("bfiP", C.c_void_p), ("wdT", C.c_void_p), ("bd", C.c_void_p),
                ("precision", C.c_int)]


_lib = None


def _hip_runtimes_mapped():
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return []


def load():
    """Load (once) and return the ctypes library handle."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "fnssl: %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C fn-ssl_amd/csrc`); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError("fnssl: %s lacks symbols %s" % (LIB_PATH, missing))
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise RuntimeError("fnssl: two HIP runtimes are mapped (%s); streams cannot be shared" % rts)

    vp, i, ll, sz, f = C.c_void_p, C.c_int, C.c_longlong, C.c_size_t, C.c_float
    lib.fnssl_abi_version.restype = i
    lib.fnssl_last_error.restype = C.c_char_p
    lib.fnssl_tuning_set.argtypes = [C.POINTER(Tuning)]
    lib.fnssl_tuning_get.argtypes = [C.POINTER(Tuning)]
    lib.fnssl_tuning_name.argtypes = [i]
    lib.fnssl_tuning_name.restype = C.c_char_p
    lib.fnssl_occupy_cus.argtypes = [i, i, vp, i, vp]
    lib.fnssl_num_frames.argtypes = [i]
    lib.fnssl_num_pairs.argtypes = [i, i]
    lib.fnssl_stft.argtypes = [vp, i, i, i, ll, ll, ll, vp, vp, vp]
    lib.fnssl_num_frames_ex.argtypes = [i, i, i]
    lib.fnssl_stft_ex.argtypes = [vp, i, i, i, ll, ll, ll, i, i, vp, vp, vp]
    lib.fnssl_array_frontend.argtypes = [vp, i, i, i, ll, ll, ll, i, i, vp, vp, f, vp, vp, vp, vp]
    lib.fnssl_forgetting_coefs.argtypes = [i, i, vp, vp]
    lib.fnssl_pair_features.argtypes = [vp, vp, vp, vp, i, i, i, i, f, vp, vp, i, vp]
    lib.fnssl_nchw_to_seq.argtypes = [vp, i, i, i, i, vp, vp]
    lib.fnssl_lstm_packed_floats.argtypes = [i, i, i]
    lib.fnssl_lstm_packed_floats.restype = sz
    lib.fnssl_lstm_pack.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.fnssl_lstm_workspace_bytes.argtypes = [i, i, i]
    lib.fnssl_lstm_workspace_bytes.restype = sz
    lib.fnssl_lstm_workspace_bytes_ex.argtypes = [i, i, i, i]
    lib.fnssl_lstm_workspace_bytes_ex.restype = sz
    lib.fnssl_lstm_forward.argtypes = [C.POINTER(LstmDesc), vp]
    lib.fnssl_head.argtypes = [vp, i, i, i, vp, vp, vp, vp]
    lib.fnssl_linear.argtypes = [vp, i, i, vp, vp, i, vp, vp]
    lib.fnssl_ipd2doa.argtypes = [vp, ll, ll, ll, ll, vp, i, i, i, i, i, i, i, vp, vp, vp, vp]
    lib.fnssl_doa_peaks.argtypes = [vp, i, i, i, i, vp, vp, vp, vp]
    lib.fnssl_dpipd_targets.argtypes = [vp, vp, i, i, i, i, vp, i, i, i, i, i, f, f, i, vp, vp, vp]
    lib.fnssl_lstm_packed_floats_bf16.argtypes = [i, i, i]
    lib.fnssl_lstm_packed_floats_bf16.restype = sz
    lib.fnssl_lstm_pack_bf16.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.fnssl_lstm_packed_floats_bf16w.argtypes = [i, i, i]
    lib.fnssl_lstm_packed_floats_bf16w.restype = sz
    lib.fnssl_lstm_pack_bf16w.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.fnssl_lstm_reserve_bytes.argtypes = [i, i, i, i]
    lib.fnssl_lstm_reserve_bytes.restype = sz
    lib.fnssl_lstm_bwd_packed_floats.argtypes = [i, i]
    lib.fnssl_lstm_bwd_packed_floats.restype = sz
    lib.fnssl_lstm_pack_bwd.argtypes = [vp, vp, i, i, i, vp]
    lib.fnssl_lstm_bwd_workspace_bytes.argtypes = [i, i, i]
    lib.fnssl_lstm_bwd_workspace_bytes.restype = sz
    lib.fnssl_lstm_backward.argtypes = [C.POINTER(LstmBwdDesc), vp]
    lib.fnssl_lstm_backward_plan.argtypes = [C.POINTER(LstmBwdDesc), C.POINTER(C.c_int)]
    lib.fnssl_lstm_backward_status.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_uint)]
    lib.fnssl_lstm_plan_rounds.argtypes = [i, i, i, i, vp, i]
    lib.fnssl_lstm_plan.argtypes = [C.POINTER(LstmDesc), vp, vp]
    lib.fnssl_mfma_f32_peak.argtypes = [vp, sz, i, i, vp, vp]
    lib.fnssl_mfma_f32_peak_clocks.argtypes = [vp, sz, i, i, vp, vp, sz, vp]
    lib.fnssl_lstm_cluster_status.argtypes = [vp, sz, i, i, i, vp, vp]
    lib.fnssl_lstm_weight_grads_workspace_bytes.argtypes = [ll, i, i, i, i]
    lib.fnssl_lstm_weight_grads_workspace_bytes.restype = sz
    lib.fnssl_lstm_weight_grads.argtypes = [C.POINTER(WgradDesc), vp]
    lib.fnssl_train_combine.argtypes = [vp, ll, ll, ll, i, i, i, i, C.POINTER(BtfView), i, C.POINTER(BtfView), i, i,
                                        C.c_uint, ll, vp]
    lib.fnssl_dropout_scale.argtypes = [vp, ll, C.c_uint, ll, vp]
    lib.fnssl_head_backward_workspace_bytes.argtypes = []
    lib.fnssl_head_backward_workspace_bytes.restype = sz
    lib.fnssl_head_backward.argtypes = [vp, vp, vp, vp, i, i, i, vp, vp, vp, i, vp, sz, vp]
    lib.fnssl_mse_loss.argtypes = [vp, vp, i, i, i, i, ll, vp, vp, i, vp, sz, vp]
    lib.fnssl_adam_step.argtypes = [vp, vp, vp, vp, ll, f, f, f, f, i, f, vp]
    lib.fnssl_array_features.argtypes = [vp, vp, vp, vp, i, i, i, C.c_float, vp, vp, i, vp]
    lib.fnssl_conv3x3_packed_floats.argtypes = [i, i, i]
    lib.fnssl_conv3x3_packed_floats.restype = sz
    lib.fnssl_conv3x3_pack.argtypes = [vp, i, i, i, vp]
    lib.fnssl_conv3x3_causal.argtypes = [vp, ll, ll, ll, i, vp, ll, ll, ll, i, vp, i, i, i, i, i, vp, i, vp]
    lib.fnssl_conv3x3_packed_floats_bf16.argtypes = [i, i, i]
    lib.fnssl_conv3x3_packed_floats_bf16.restype = sz
    lib.fnssl_conv3x3_pack_bf16.argtypes = [vp, i, i, i, vp]
    lib.fnssl_conv3x3_causal_bf16.argtypes = [vp, ll, ll, ll, i, vp, ll, ll, ll, i, vp, i, i, i, i, i, vp, i, vp]
    lib.fnssl_conv3x3_causal_bf16a.argtypes = [vp, ll, ll, ll, i, vp, ll, ll, ll, i, vp, i, i, i, i, i, vp, i, vp]
    lib.fnssl_avgpool_time.argtypes = [vp, i, i, i, i, vp, vp]
    lib.fnssl_conv3x3_packed_bytes_bf16x.argtypes = [i, i, i]
    lib.fnssl_conv3x3_packed_bytes_bf16x.restype = sz
    lib.fnssl_conv3x3_pack_bf16x.argtypes = [vp, i, i, i, vp]
    lib.fnssl_conv3x3_causal_bf16x.argtypes = [vp, ll, ll, ll, i, vp, ll, ll, ll, i, vp, i, i, i, i, i, i, i, vp, i, vp]
    lib.fnssl_avgpool_time_bf16.argtypes = [vp, i, i, i, i, vp, vp]
    lib.fnssl_forward_workspace_bytes.argtypes = [i, i, i, i, i]
    lib.fnssl_forward_workspace_bytes.restype = sz
    lib.fnssl_forward.argtypes = [C.POINTER(Net), vp, i, i, i, vp, vp, sz, i, vp]
    lib.fnssl_train_create.argtypes = [i, C.POINTER(vp)]
    lib.fnssl_train_destroy.argtypes = [vp]
    lib.fnssl_train_destroy.restype = None
    lib.fnssl_train_param_floats.argtypes = [vp]
    lib.fnssl_train_param_floats.restype = ll
    lib.fnssl_train_param_offset.argtypes = [vp, i, i, i]
    lib.fnssl_train_param_offset.restype = ll
    lib.fnssl_train_map_bytes.argtypes = [vp]
    lib.fnssl_train_map_bytes.restype = sz
    lib.fnssl_train_upload_maps.argtypes = [vp, vp, vp]
    lib.fnssl_train_workspace_bytes.argtypes = [vp, i, i, i]
    lib.fnssl_train_workspace_bytes.restype = sz
    lib.fnssl_train_backward.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, C.c_uint, ll, ll, vp, vp, sz, vp]
    lib.fnssl_train_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, C.c_uint, f, f, f, f, i, vp, vp, sz, vp]
    PV = C.POINTER(BtfView)
    lib.fnssl_sn_layernorm.argtypes = [vp, ll, i, vp, vp, f, vp, vp]
    lib.fnssl_sn_encoder.argtypes = [vp, ll, ll, ll, ll, i, i, i, i, vp, vp, vp, vp, vp, ll, ll, ll, i, vp]
    lib.fnssl_sn_fconv.argtypes = [PV, i, i, i, C.POINTER(SnFconvW), i, i, vp, ll, ll, ll, i, vp]
    lib.fnssl_sn_full.argtypes = [PV, i, i, i, C.POINTER(SnFullW), i, vp, ll, ll, ll, i, vp]
    lib.fnssl_sn_mamba_workspace_bytes.argtypes = [i, i, i]
    lib.fnssl_sn_mamba_workspace_bytes.restype = sz
    lib.fnssl_sn_mamba.argtypes = [PV, i, i, i, C.POINTER(SnMambaW), i, i, vp, vp, i, vp, ll, ll, ll, vp, sz, i, vp]
    lib.fnssl_sn_head.argtypes = [PV, i, i, i, vp, vp, vp, vp, vp, vp]
    lib.fnssl_sn_forward_workspace_bytes.argtypes = [i, i, i]
    lib.fnssl_sn_forward_workspace_bytes.restype = sz
    lib.fnssl_sn_state_floats.argtypes = [C.POINTER(SnNet), i, i]
    lib.fnssl_sn_state_floats.restype = sz
    lib.fnssl_sn_forward.argtypes = [C.POINTER(SnNet), vp, ll, ll, ll, ll, i, i, i, vp, i, vp, vp, sz, vp]
    lib.fnssl_timing_enable.argtypes = [i]
    lib.fnssl_timing_select.argtypes = [C.c_char_p]
    lib.fnssl_timing_collect.argtypes = [i, vp, vp, vp, vp]
    if lib.fnssl_abi_version() != ABI_VERSION:
        raise RuntimeError("fnssl: ABI version mismatch (library %d, binding %d)"
                           % (lib.fnssl_abi_version(), ABI_VERSION))
    _lib = lib
    refresh_tuning()
    return lib


# --------------------------------------------------------------------------- #
# tuning: FNSSL_<KNOB> environment variables are a PYTHON-side convenience
# --------------------------------------------------------------------------- #
_tune_index = None
# knobs the old getenv()-style switches treated as "set = on" whatever the value
_PRESENCE = {"NO_F32_CLUSTER", "NO_F32C_B1", "TRAIN_NO_F32_CLUSTER", "F32C_NO_ROTATE", "NO_CLUSTER", "NO_CLUSTER_B1",
             "NO_CLUSTER_H128", "CLUSTER_SPREAD", "BF16W_SOLO", "SN_SCALAR", "STFT_PER_FRAME"}
# knobs whose environment spelling differs from FNSSL_<name>
_ENV_ALIAS = {"LSTM_NO_STATIC": "FNSSL_LSTM_NO_STATIC"}


def tuning_names():
    """knob name -> index, as the loaded library reports them (fnssl_tuning_name)."""
    global _tune_index
    if _tune_index is None:
        lib = _lib
        if lib is None:
            raise RuntimeError("fnssl: load() first")
        idx, k = {}, 0
        while True:
            n = lib.fnssl_tuning_name(k)
            if n is None:
                break
            idx[n.decode()] = k
            k += 1
        _tune_index = idx
    return _tune_index


def tuning_from_env(environ=None) -> Tuning:
    """The Tuning an environment describes: FNSSL_<KNOB>=<int> (flags: any value / "1")."""
    environ = os.environ if environ is None else environ
    t = Tuning()
    t.struct_bytes = C.sizeof(Tuning)
    for name, k in tuning_names().items():
        v = environ.get(_ENV_ALIAS.get(name, "FNSSL_" + name))
        if v is None:
            continue
        try:
            iv = int(v)
        except ValueError:
            iv = 1
        if name in _PRESENCE:
            iv = 1
        if name == "CLUSTER_TEST_STALL":
            iv = iv + 1 if iv >= 0 else 0          # the knob stores member + 1 (0 = no fault injection)
        t.knob[k] = iv
    return t


def refresh_tuning(environ=None):
    """Re-read the FNSSL_* environment variables into the library's default tuning (fnssl_tuning_set)."""
    if _lib is None:
        return
    t = tuning_from_env(environ)
    if _lib.fnssl_tuning_set(C.byref(t)) != 0:
        raise RuntimeError("fnssl: tuning_set failed: %s" % _lib.fnssl_last_error().decode("utf-8", "replace"))


def make_tuning(base: Tuning = None, **knobs) -> Tuning:
    """A Tuning with the given knobs (lower- or upper-case names) on top of ``base`` (default: the thread's current)."""
    load()
    t = Tuning()
    if base is None:
        _lib.fnssl_tuning_get(C.byref(t))
    else:
        C.memmove(C.byref(t), C.byref(base), C.sizeof(Tuning))
    t.struct_bytes = C.sizeof(Tuning)
    names = tuning_names()
    for k, v in knobs.items():
        if k.upper() not in names:
            raise KeyError("fnssl: unknown tuning knob %r (known: %s)" % (k, ", ".join(sorted(names))))
        t.knob[names[k.upper()]] = int(v)
    return t


class tuning:
    """``with _lib.tuning(no_static3=1, cluster_spin_limit=20000): ...`` — the calls inside the block use
    these knobs on top of the current ones; the previous tuning is restored on exit.

    PROCESS-WIDE and not thread-safe: it swaps the library's default tuning (``fnssl_tuning_set``), so calls issued by other
    threads / on other streams during the block see the knobs too, and a backward that autograd runs AFTER the block (on its
    worker thread) does not.  For per-call knobs pass a ``Tuning`` (``make_tuning(...)``) through the op's ``tuning=`` argument
    (``ops.lstm_layer``, ``ops.lstm_backward``) or ``TrainGraph(bwd_tuning=...)``: that travels in the descriptor of the call."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        load()
        self.prev = Tuning()
        _lib.fnssl_tuning_get(C.byref(self.prev))
        t = make_tuning(self.prev, **self.knobs)
        check(_lib.fnssl_tuning_set(C.byref(t)), "tuning_set")
        return t

    def __exit__(self, *exc):
        _lib.fnssl_tuning_set(C.byref(self.prev))
        return False


def check(rc: int, what: str = ""):
    """Raise RuntimeError (like ATen does for shape errors) on a non-zero status."""
    if rc != 0:
        msg = load().fnssl_last_error().decode("utf-8", "replace")
        raise RuntimeError("fnssl %s failed (%d): %s" % (what, rc, msg))
