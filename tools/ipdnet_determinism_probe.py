"""Stage-by-stage determinism probe of the bf16 IPDnet forward at config 3's batch (one stream): each stage is run
several times on the same input, with the freed memory poisoned in between (so a read of uninitialised scratch shows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from fnssl import weights as W, ops
import importlib.util
spec = importlib.util.spec_from_file_location("dropin", os.path.join(ROOT, "fn-ssl_amd", "IPDnet", "FixedAarryIPDnet.py"))
M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
dev = torch.device("cuda:0")
nb, nf, nt = int(os.environ.get("NB", 64)), 256, int(os.environ.get("NT", 300))
POISON = os.environ.get("POISON", "1") == "1"
os.environ["FNSSL_IPDNET_ONE_STREAM"] = "1"
(lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
sd = W.make_ipdnet_state(4500, 16, 256, 2, True)
net = M.IPDnet(input_size=16, hidden_size=256, max_track=2, is_online=True).eval()
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
net.to(dev).bfloat16()
x = torch.randn((nb, 16, nf, nt), device=dev).bfloat16().float()
xs = ops.nchw_to_seq(x)
xp = M._pad_channels(xs, 16)

def poison():
    if not POISON: return
    t = torch.full((int(40e9) // 4,), float("nan"), device=dev)   # 40 GB of NaN handed back to the caching allocator
    del t

def rep(name, fn, n=4):
    outs = []
    for i in range(n):
        poison()
        ops.release_workspaces()
        o = fn()
        torch.cuda.synchronize()
        outs.append(o.float().clone())
    eq = [bool(torch.equal(outs[0], o)) for o in outs]
    nan = [bool(torch.isnan(o).any()) for o in outs]
    d = [float((o - outs[0]).abs().max()) for o in outs]
    print("%-28s equal to run 0: %s  nan: %s  max diff %s" % (name, eq, nan, ["%.2g" % v for v in d]), flush=True)
    return outs[0]

y1 = rep("block_1", lambda: net.block_1.run(None, xp))
y1 = net.block_1.run(None, xp)
y2 = rep("block_2", lambda: net.block_2.run(y1, xp))
y2 = net.block_2.run(y1, xp)
c = rep("conv", lambda: net.conv.run(y2.permute(0, 2, 1, 3), xp.permute(0, 2, 1, 3)))
o = rep("whole forward", lambda: net(x))

if os.environ.get("LAYERS"):
    blk = net.block_1
    full_w, narr_w = blk._streams(dev)
    def run_full():
        f = torch.empty((nb, nt, nf, 256), dtype=torch.bfloat16, device=dev)
        ops.lstm_layer("full", xp, None, None, full_w, 128, f, bf16=True, wide=True)
        return f
    f0 = rep("block_1 full-band only", run_full, 6)
    fx = run_full()
    def run_narr():
        n = torch.empty((nb, nf, nt, 256), dtype=torch.bfloat16, device=dev)
        ops.lstm_layer("narrow", fx, None, xp, narr_w, 256, n.permute(0, 2, 1, 3), bf16=True, wide=True)
        return n
    rep("block_1 narrow-band only", run_narr, 6)
    # which sequences differ between two runs of the full-band layer
    a, b = run_full().float(), run_full().float()
    d = (a != b).reshape(nb * nt, nf, 2, 128)
    seqs = torch.nonzero(d.any(3).any(1))           # (sequence, direction)
    print("differing (sequence, dir):", seqs[:20].tolist(), "count", len(seqs))
    if len(seqs):
        s0, d0 = seqs[0].tolist()
        st = torch.nonzero(d[s0, :, d0].any(1)).flatten()
        print("first differing step of seq %d dir %d: %s .. %s; units at that step: %s" % (s0, d0, int(st[0]), int(st[-1]),
              torch.nonzero(d[s0, int(st[0]), d0]).flatten().tolist()[:40]))
        grp = sorted({int(s) // 32 for s, _ in seqs.tolist()})
        print("groups (32 seq):", grp[:40])

if os.environ.get("LAYERS"):
    import time
    for _ in range(3): run_full()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run_full()
    torch.cuda.synchronize(); print("block_1 full-band layer: %.3f ms" % ((time.perf_counter() - t0) * 100))
