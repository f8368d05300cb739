#!/usr/bin/env python3
"""Round 4: per-workgroup timeline of lstm_f32c_kernel at config 2's full-band layer (ablate build, FNSSL_F32C_ABL=512:
every workgroup's waves 0 and 15 print their start / end on the 100 MHz counter and their shader cycles).  Summarises
start and end stagger, lifetime and clock per XCD (block & 7)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
    import torch
    from fnssl import ops, weights as W
    dev = torch.device("cuda:0")
    nb, nt, nf, H = 192, 300, 256, 128
    sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(256, H, True)], seed=1)
    w = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], 256, 0, dev)
         for s in ("", "_reverse")]
    x = torch.randn((nb, nt, nf, 256), device=dev) * 0.3
    out = torch.empty((nb, nt, nf, 256), device=dev)
    osum = torch.empty_like(out)
    fn = lambda: ops.lstm_layer("full", x, None, None, w, H, out, skip=x, out_sum=osum)
    fn(); fn()
    torch.cuda.synchronize()
    os.environ["FNSSL_F32C_ABL"] = sys.argv[2] if len(sys.argv) > 2 else "512"
    (lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    print("KERNEL_CALL_MS %.3f" % e0.elapsed_time(e1), flush=True)
    sys.exit(0)
env = dict(os.environ, FNSSL_LIB_PATH=os.path.join(ROOT, "fn-ssl_amd", "csrc", "libfnssl_hip_abl.so"))
for mask in (sys.argv[1:] or ["512"]):
    txt = subprocess.run([sys.executable, __file__, "child", mask], env=env, capture_output=True, text=True).stdout
    rows = [tuple(float(v) for v in m.groups()) for m in re.finditer(r"f32c blk (\d+) wave (\d+) start (\d+) end (\d+) cycles (\d+) MHz ([\d.]+)", txt)]
    call = re.search(r"KERNEL_CALL_MS ([\d.]+)", txt)
    print("mask %s: %d rows, whole call %s ms" % (mask, len(rows), call.group(1) if call else "?"))
    if not rows:
        print(txt[-2000:]); continue
    t0 = min(r[2] for r in rows)
    t1 = max(r[3] for r in rows)
    print("  first start -> last end: %.3f ms; starts spread %.3f ms; ends spread %.3f ms" %
          ((t1 - t0) * 1e-5, (max(r[2] for r in rows) - t0) * 1e-5, (t1 - min(r[3] for r in rows)) * 1e-5))
    for xcd in range(8):
        rs = [r for r in rows if int(r[0]) % 8 == xcd]
        if rs:
            print("  XCD %d: %3d waves  start %.3f..%.3f ms  end %.3f..%.3f ms  life %.2f..%.2f ms  cycles %.1f..%.1f M  clock %.0f..%.0f MHz" %
                  (xcd, len(rs), (min(r[2] for r in rs) - t0) * 1e-5, (max(r[2] for r in rs) - t0) * 1e-5,
                   (min(r[3] for r in rs) - t0) * 1e-5, (max(r[3] for r in rs) - t0) * 1e-5,
                   min(r[3] - r[2] for r in rs) * 1e-5, max(r[3] - r[2] for r in rs) * 1e-5,
                   min(r[4] for r in rs) * 1e-6, max(r[4] for r in rs) * 1e-6, min(r[5] for r in rs), max(r[5] for r in rs)))
    for wv in (0, 15):
        rs = [r for r in rows if int(r[1]) == wv]
        print("  wave %2d: cycles %.1f..%.1f M, life %.2f..%.2f ms" % (wv, min(r[4] for r in rs) * 1e-6, max(r[4] for r in rs) * 1e-6,
                                                                      min(r[3] - r[2] for r in rs) * 1e-5, max(r[3] - r[2] for r in rs) * 1e-5))
