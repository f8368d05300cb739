"""Times IPDnet2's front end + encoder with the features contiguous ([B, 30, 256, T]) or frame-major (a permuted view)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fn-ssl_amd"), ROOT]
import torch
from fnssl import ops
sys.argv = ["bench.py"]
import bench
import argparse
a = argparse.Namespace(nb=0, frames=300, fp32=False, features_in=False)
wl = bench.Ipdnet2Forward(a, torch.device("cuda:0"), 0, 1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("front end contiguous   %.3f ms" % t(lambda: ops.preprocess_ipdnet2(wl.sig)))
print("front end frame-major  %.3f ms" % t(lambda: ops.preprocess_ipdnet2(wl.sig, frame_major=True)))
xc = ops.preprocess_ipdnet2(wl.sig); xf = ops.preprocess_ipdnet2(wl.sig, frame_major=True)
print("net on contiguous      %.3f ms" % t(lambda: wl.net(xc)))
print("net on frame-major     %.3f ms" % t(lambda: wl.net(xf)))
print("equal outputs:", torch.equal(wl.net(xc), wl.net(xf)))
print("whole step             %.3f ms" % t(wl.step))
ops.timing_select(None); ops.timing_enable(True); wl.step(); torch.cuda.synchronize(); ops.timing_enable(False)
for k, v in sorted(ops.timing_collect().items()): print("  %-22s %.3f ms x%d" % (k, v["ms"], v["count"]))
