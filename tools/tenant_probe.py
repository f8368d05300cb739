#!/usr/bin/env python3
"""What a CU-holding tenant (fnssl_occupy_cus: the stand-in for RCCL's persistent kernels) does to the cluster-resident BPTT
kernel: time and status word of block 1's layer (clusters of 2 on all 256 CUs by default) with / without the tenant and
with / without fnssl_tuning RESERVED_CUS."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "fn-ssl_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

from fnssl import _lib, ops  # noqa: E402
from fnssl import weights as W  # noqa: E402

dev = torch.device("cuda:0")
H, c0g, nb, nt, nf = 128, 0, 28, 300, 64
c_in = 16
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c_in, H, True)], seed=1)
sfx = ("", "_reverse")
packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c_in, 0, dev) for s in sfx]
bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
g = torch.Generator(device=dev)
g.manual_seed(6)
x = torch.randn((nb, nt, nf, c_in), generator=g, device=dev) * 0.7
dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
out = torch.empty((nb, nt, nf, 2 * H), device=dev)
reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)
da = torch.empty((nb, nt, nf, 8 * H), device=dev)
side = torch.cuda.Stream(device=dev)
stop = torch.zeros(17, dtype=torch.int32).pin_memory()


def run(tenant, **knobs):
    with _lib.tuning(**knobs):
        fam = ops.lstm_backward("full", reserve, dh, da, None, bw, H, 0, plan_only=True)
        torch.cuda.synchronize()
        if tenant:
            stop.zero_()
            ops.occupy_cus(16, stop, max_ms=2000, stream=side)
            t_end = time.time() + 1.0
            while int(stop[1:].sum()) < 16 and time.time() < t_end:
                time.sleep(0.001)
        ops.cluster_fallbacks(dev, reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _, _, word = ops.lstm_backward("full", reserve, dh, da, None, bw, H, 0, status=True)
        e1.record()
        e1.synchronize()
        res = int(stop[1:].sum())
        stop[0] = 1
        side.synchronize()
        print("tenant %d knobs %s: family %s, %.2f ms, status 0x%x, fallbacks %d, tenant workgroups resident %d"
              % (tenant, knobs, fam, e0.elapsed_time(e1), word, ops.cluster_fallbacks(dev), res), flush=True)


run(False)
run(False)
run(True)
run(True, cluster_spin_limit=2000)
run(True, cluster_spin_limit=200)
run(True, reserved_cus=16)
run(False, reserved_cus=16)
