"""Does replaying the IPDnet2 forward (106 launches) as a HIP graph shorten the gaps between its small kernels?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fn-ssl_amd"))
import numpy as np, torch
from fnssl import weights as W
from IPDnet2 import IPDnet2 as M
dev = torch.device("cuda:0")
sd = W.make_ipdnet2_state(1, dim_input=30, num_layers=8)
net = M.OnlineSpatialNet(dim_input=30, dim_output=16, num_layers=8, dim_hidden=96, dim_squeeze=8, num_freqs=256,
                         attention="mamba(16,4)", rope=False).eval()
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
net.to(dev)
x = torch.randn((64, 30, 256, 250), device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    ref = net(x)
    print("eager  %.3f ms" % timeit(lambda: net(x)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): net(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = net(x)
    print("graph  %.3f ms" % timeit(lambda: g.replay()))
    g.replay(); torch.cuda.synchronize()
    print("equal", torch.equal(out, ref))
