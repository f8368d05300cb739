#!/usr/bin/env python3
"""Round 4: bottom-up / top-down timing ablations of the two config-2 kernels on the ABLATE build of the library
(`make ABLATE=1` -> csrc/libfnssl_hip_abl.so, loaded through FNSSL_LIB_PATH; results of ablated runs are wrong by
construction).  One layer at config 2's size per line: wall time by HIP events, fraction of the fp32 MFMA roof.

    FNSSL_LIB_PATH=fn-ssl_amd/csrc/libfnssl_hip_abl.so python tools/ablate_r04.py [f32c] [f32c_b1] [static2]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch
from fnssl import ops, weights as W

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
nb, nt, nf = 192, 300, 256


def packed(c0, c2, H, bidir, seed=1):
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c0 + c2, H, bidir)], seed=seed)
    return [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s],
                          c0, c2, dev) for s in ([""] + (["_reverse"] if bidir else []))]


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def sweep(name, env, masks, fn, flops):
    for m in masks:
        if m:
            os.environ[env] = str(m)
        else:
            os.environ.pop(env, None)
        ms = timed(fn)
        print("%-8s %s=%-4d %8.2f ms  %6.1f TFLOP/s  %.3f of 157.3" % (name, env, m, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3), flush=True)
    os.environ.pop(env, None)


def prio_sweep(name, env, modes, fn, flops):
    for m in modes:
        if m:
            os.environ[env] = str(m)
        else:
            os.environ.pop(env, None)
        ms = [timed(fn) for _ in range(2)]
        print("%-8s %s=%d  %s ms  %.3f of 157.3" % (name, env, m, " ".join("%.2f" % v for v in ms), flops / min(ms) / 1e9 / 157.3), flush=True)
    os.environ.pop(env, None)


def env_sweep(name, env, values, fn, flops):
    for v in values:
        os.environ[env] = v
        ms = [timed(fn) for _ in range(2)]
        print("%-8s %s=%s  %s ms  %.3f of 157.3" % (name, env, v, " ".join("%.2f" % t for t in ms), flops / min(ms) / 1e9 / 157.3), flush=True)
    os.environ.pop(env, None)


which = sys.argv[1:] or ["peak", "f32c", "f32c_b1", "static2"]
ABL_LIB = "abl" in os.environ.get("FNSSL_LIB_PATH", "")
if "peak" in which:
    for wps in (1, 2, 4):
        print("mfma_f32_peak  %d waves per SIMD: %.1f TFLOP/s" % (wps, ops.mfma_f32_peak(waves_per_simd=wps)), flush=True)
g = torch.Generator(device=dev)
g.manual_seed(0)

if "f32c" in which:
    # bits: 1 one group's addressing, 2 cheap gates, 4 no tag waits, 8 no x loads, 16 no h loads, 32 no stores,
    #       64 no c / skip loads, 128 no LDS record reads, 256 no tag loads / publishes
    x = torch.randn((nb, nt, nf, 256), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 256), device=dev)
    osum = torch.empty_like(out)
    w = packed(256, 0, 128, True)
    fl = 2.0 * 4 * 128 * (256 + 128) * nb * nt * nf * 2
    fn = lambda: ops.lstm_layer("full", x, None, None, w, 128, out, skip=x, out_sum=osum)
    for _ in range(2):
        os.environ.pop("FNSSL_F32C_NO_ROTATE", None)
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
        print("f32c rotate     %s" % " ".join("%.2f" % timed(fn) for _ in range(2)), flush=True)
        os.environ["FNSSL_F32C_NO_ROTATE"] = "1"
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
        print("f32c no-rotate  %s" % " ".join("%.2f" % timed(fn) for _ in range(2)), flush=True)
    os.environ.pop("FNSSL_F32C_NO_ROTATE", None)
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    if ABL_LIB:
        M = 8 | 16 | 64                  # every load except the tags
        sweep("f32c", "FNSSL_F32C_ABL", [512, 512 | 511, 512 | M | 32 | 4 | 256, 1, 511 - 1, 0], fn, fl)
    os.environ["FNSSL_NO_F32_CLUSTER"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    print("rounds   %8.2f ms" % timed(fn), flush=True)
    os.environ.pop("FNSSL_NO_F32_CLUSTER")
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    del x, out, osum
    torch.cuda.empty_cache()

if "f32c_b1" in which:
    x = torch.randn((nb, nt, nf, 4), generator=g, device=dev)
    out = torch.empty((nb, nt, nf, 256), device=dev)
    w = packed(4, 0, 128, True)
    fl = 2.0 * 4 * 128 * (4 + 128) * nb * nt * nf * 2
    fn = lambda: ops.lstm_layer("full", x, None, None, w, 128, out)
    for _ in range(2):
        os.environ.pop("FNSSL_F32C_NO_ROTATE", None)
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
        print("b1 rotate     %s" % " ".join("%.2f" % timed(fn) for _ in range(2)), flush=True)
        os.environ["FNSSL_F32C_NO_ROTATE"] = "1"
        (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
        print("b1 no-rotate  %s" % " ".join("%.2f" % timed(fn) for _ in range(2)), flush=True)
    os.environ.pop("FNSSL_F32C_NO_ROTATE", None)
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    if ABL_LIB:
        sweep("f32c_b1", "FNSSL_F32C_ABL", [512, 1, 2, 511, 0], fn, fl)
    del x, out
    torch.cuda.empty_cache()

if "static2" in which:
    # bits: 1 no x loads, 2 cheap gates, 4 no stores, 8 no ring barrier, 16 no c / skip loads, 32 no h reload,
    #       64 no LDS record reads, 128 no weight staging
    os.environ["FNSSL_ABL_STATIC2"] = "1"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    F = torch.randn((nb, nt, nf, 256), generator=g, device=dev) * 0.3
    out = torch.empty((nb, nf, nt, 256), device=dev).permute(0, 2, 1, 3)
    osum = torch.empty((nb, nf, nt, 256), device=dev).permute(0, 2, 1, 3)
    w = packed(256, 0, 256, False)
    fl = 2.0 * 4 * 256 * (256 + 256) * nb * nt * nf
    fn = lambda: ops.lstm_layer("narrow", F, None, None, w, 256, out, skip=F, out_sum=osum)
    env_sweep("narrow/sum", "FNSSL_STATIC_PRIO", ["0", "7", "8", "0", "7", "8"], fn, fl)
    fn0 = lambda: ops.lstm_layer("narrow", F, None, None, w, 256, out)
    env_sweep("narrow/plain", "FNSSL_NO_STATIC3", ["0", "1", "0", "1"], fn0, fl)
    env_sweep("narrow/plain", "FNSSL_STATIC3_CHQ", ["9", "6"], fn0, fl)
    X2 = torch.randn((nb, nt, nf, 4), generator=g, device=dev)
    w1 = packed(256, 4, 256, False)
    fl1 = 2.0 * 4 * 256 * (260 + 256) * nb * nt * nf
    fn1 = lambda: ops.lstm_layer("narrow", F, None, X2, w1, 256, out, skip=F, out_sum=osum)
    env_sweep("narrow/b1", "FNSSL_NO_STATIC3", ["0", "1", "0", "1"], fn1, fl1)
    env_sweep("narrow/b1", "FNSSL_STATIC3_CHQ", ["9", "6"], fn1, fl1)
    if ABL_LIB:
        pass
