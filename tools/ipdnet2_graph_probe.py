#!/usr/bin/env python3
"""Is the IPDnet2 forward (config 5, bf16) bound by launch gaps?  Eager launches against a hipGraph replay of the same
forward (torch.cuda.graph capture of the library's launches on the capturing stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "fn-ssl_amd")); sys.path.insert(0, ROOT)
import torch
from ipdnet2_bench import build
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
sd, net = build(dev, 15, 8)
if os.environ.get("FP32") != "1":
    net = net.bfloat16()
x = torch.randn((64, 30, 256, 250), device=dev)


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


y0 = net(x); torch.cuda.synchronize()
print("eager  %.3f ms" % timed(lambda: net(x)), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): net(x)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = net(x)
torch.cuda.synchronize()
print("graph  %.3f ms" % timed(g.replay), flush=True)
print("same output:", bool(torch.equal(y, y0)))
