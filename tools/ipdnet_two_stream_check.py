"""Repeats the bf16 IPDnet forward at config 3's batch on the two-stream path and reports which runs / utterances differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fn-ssl_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from fnssl import weights as W
import importlib.util
spec = importlib.util.spec_from_file_location("dropin", os.path.join(ROOT, "fn-ssl_amd", "IPDnet", "FixedAarryIPDnet.py"))
M = importlib.util.module_from_spec(spec); spec.loader.exec_module(M)
dev = torch.device("cuda:0")
nb, nf, nt = int(os.environ.get("NB", 64)), 256, int(os.environ.get("NT", 300))
sd = W.make_ipdnet_state(4500, 16, 256, 2, True)
net = M.IPDnet(input_size=16, hidden_size=256, max_track=2, is_online=True).eval()
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
net.to(dev).bfloat16()
x = torch.randn((nb, 16, nf, nt), device=dev).bfloat16().float()
os.environ["FNSSL_IPDNET_ONE_STREAM"] = "1"
(lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
ones = [net(x) for _ in range(3)]
torch.cuda.synchronize()
print("one-stream runs equal:", [bool(torch.equal(ones[0], o)) for o in ones])
del os.environ["FNSSL_IPDNET_ONE_STREAM"]
(lambda m: m and m.refresh_tuning())(__import__("sys").modules.get("fnssl._lib"))   # FNSSL_* knobs are parsed by fnssl/_lib.py
os.environ.setdefault("FNSSL_IPDNET_STREAMS", "2")      # part-batches on separate streams are opt-in since the cluster kernels
outs = [net(x) for _ in range(8)]
torch.cuda.synchronize()
for i, o in enumerate(outs):
    d = (o != ones[0])
    per_utt = d.reshape(nb, -1).sum(1)
    bad = [(int(b), int(per_utt[b])) for b in torch.nonzero(per_utt).flatten().tolist()]
    print("two-stream run %d: %d differing values, max abs diff %.3g, utterances %s" % (i, int(d.sum()), float((o - ones[0]).abs().max()), bad[:12]))
