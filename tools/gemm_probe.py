#!/usr/bin/env python3
"""Which library path is fastest for the weight-gradient GEMMs  dW[4H, C] = dA^T[4H, R] @ X[R, C]  (R = millions of
rows)?  Plain addmm vs manual split-K through bmm."""
import sys, time
import torch
dev = torch.device("cuda:0")
R = 32 * 256 * 300
for (M, C, ld) in [(1024, 256, 1024), (512, 256, 1024), (1024, 256, 1024)]:
    da = torch.randn((R, ld), device=dev)[:, :M]
    x = torch.randn((R, C), device=dev)
    out = torch.zeros((M, C), device=dev)
    def t(fn, n=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    flops = 2.0 * M * C * R
    a = t(lambda: out.addmm_(da.t(), x))
    res = ["addmm %.2f ms %.0f TF" % (a * 1e3, flops / a / 1e12)]
    b2 = t(lambda: torch.mm(x.t(), da))      # transposed problem [C, R] x [R, M]
    res.append("mm(x^T, dA) %.2f ms %.0f TF" % (b2 * 1e3, flops / b2 / 1e12))
    for S in (16, 64, 256):
        dav = da.unflatten(0, (S, R // S))          # [S, R/S, M] (strided)
        xv = x.unflatten(0, (S, R // S))
        c = t(lambda: torch.bmm(dav.transpose(1, 2), xv).sum(0))
        res.append("bmm S=%d %.2f ms %.0f TF" % (S, c * 1e3, flops / c / 1e12))
        d = t(lambda: torch.bmm(xv.transpose(1, 2), dav).sum(0))
        res.append("bmmT S=%d %.2f ms %.0f TF" % (S, d * 1e3, flops / d / 1e12))
    print((M, C, ld), " | ".join(res), flush=True)
