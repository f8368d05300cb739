import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd"))
import torch
from fnssl import ops, weights as W
dev = torch.device("cuda:0")
H = 128
c0g, nb, nt, nf = int(sys.argv[1]), 32, int(sys.argv[3]) if len(sys.argv) > 3 else 300, int(sys.argv[2])
c_in = c0g if c0g else 16
sd = W.make_state([("L." + n, s) for n, s in W.lstm_param_shapes(c_in, H, True)], seed=770)
sfx = ("", "_reverse")
packed = [ops.pack_lstm(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], sd["L.bias_ih_l0" + s], sd["L.bias_hh_l0" + s], c_in, 0, dev) for s in sfx]
bw = [torch.from_numpy(ops.pack_lstm_bwd_host(sd["L.weight_ih_l0" + s], sd["L.weight_hh_l0" + s], c0g)).to(dev) for s in sfx]
g = torch.Generator(device=dev); g.manual_seed(6)
x = torch.randn((nb, nt, nf, c_in), generator=g, device=dev) * 0.7
dh = torch.randn((nb, nt, nf, 2 * H), generator=g, device=dev) * 0.3
out = torch.empty((nb, nt, nf, 2 * H), device=dev)
reserve = torch.zeros((ops.lstm_reserve_floats(nb * nt, H, 2, nf),), device=dev)
ops.lstm_layer("full", x, None, None, packed, H, out, reserve=reserve)
def run():
    da = torch.full((nb, nt, nf, 8 * H), float("nan"), device=dev)
    dx = torch.full((nb, nt, nf, 2 * c0g), float("nan"), device=dev) if c0g else None
    _, _, word = ops.lstm_backward("full", reserve, dh, da, dx, bw, H, c0g, status=True)
    return da, dx, word
a, xa, wa = run()
os.environ["FNSSL_BWD_NO_CLUSTER"] = "1"
(lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
b, xb, wb = run()
print("status", wa, wb, "nan in a:", int(torch.isnan(a).sum()))
bad = (a != b) | torch.isnan(a)
print("mismatching elements", int(bad.sum()), "of", bad.numel())
bad4 = bad.reshape(nb * nt // 16, 16, nf, 2, 4, 8, 16)   # group, seq, step, dir, gate, slice, unit
print("by step (array index):", bad4.sum(dim=(0, 1, 3, 4, 5, 6)).tolist())
print("by dir:", bad4.sum(dim=(0, 1, 2, 4, 5, 6)).tolist())
print("by gate:", bad4.sum(dim=(0, 1, 2, 3, 5, 6)).tolist())
print("by slice:", bad4.sum(dim=(0, 1, 2, 3, 4, 6)).tolist())
pg = bad4.sum(dim=(1, 2, 3, 4, 5, 6))
print("groups with mismatches:", int((pg > 0).sum()), "of", pg.numel(), "first:", torch.nonzero(pg)[:40].flatten().tolist())
err = (a - b).abs().nan_to_num(nan=1e9)
print("max abs err", float(err.max()), "max |b|", float(b.abs().max()))
if xa is not None:
    badx = (xa != xb) | torch.isnan(xa)
    print("dx mismatches", int(badx.sum()), "of", badx.numel())
gi = int(torch.nonzero(pg)[0])
print("group", gi)
for d in range(2):
    for st in range(nf):
        t = bad4[gi, :, st, d]   # seq, gate, slice, unit
        print("dir", d, "step", st, "per gate x slice:", t.sum(dim=(0, 3)).tolist(), " per seq:", t.sum(dim=(1, 2, 3)).tolist())
t = bad4[gi, :, nf - 1, 0]
print("dir 0 first processed step, wrong (seq, gate, slice, unit):", torch.nonzero(t).tolist()[:64])
ta = a.reshape(nb * nt // 16, 16, nf, 2, 4, 8, 16)[gi, :, nf - 1, 0]
tb = b.reshape(nb * nt // 16, 16, nf, 2, 4, 8, 16)[gi, :, nf - 1, 0]
for idx in torch.nonzero(t).tolist()[:8]:
    print(idx, float(ta[tuple(idx)]), float(tb[tuple(idx)]))
