"""Times fnssl_lstm_weight_grads at BASELINE config 4's layer shapes (32 pairs x 256 bins x 300 frames)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fn-ssl_amd"), ROOT]
import torch
from fnssl import ops
dev = torch.device("cuda:0")
rows = 32 * 256 * 300
for name, H, nd, c0, c2, nsteps in (("full blk1", 128, 2, 4, 0, 256), ("full blk2/3", 128, 2, 256, 0, 256),
                                   ("narrow blk1", 256, 1, 256, 4, 300), ("narrow blk2/3", 256, 1, 256, 0, 300)):
    da = torch.randn((rows, nd * 4 * H), device=dev)
    x0 = torch.randn((rows, c0), device=dev)
    x2 = torch.randn((rows, c2), device=dev) if c2 else None
    h = torch.randn((rows, nd * H), device=dev)
    g = [[torch.zeros(s, device=dev) for _ in range(nd)] for s in ((4 * H, c0 + c2), (4 * H, H), (4 * H,), (4 * H,))]
    f = lambda: ops.lstm_weight_grads(da, x0, x2, h, H, nd, nsteps, *g)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    fl = 2.0 * rows * nd * 4 * H * (c0 + c2 + H)
    print("%-14s %.2f ms  %.1f TFLOP/s (%.0f %% of the fp32 MFMA roof)" % (name, dt * 1e3, fl / dt / 1e12, fl / dt / 157.3e10))
    del da, x0, x2, h
