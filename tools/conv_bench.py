"""Time IPDnet's conv1 (272 -> 128, [64, 256, 300]) through conv_bf16x with the ablation bits of FNSSL_CONVX_ABL."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fn-ssl_amd"))
if len(sys.argv) > 1:
    import numpy as np, torch
    from fnssl import ops
    dev = torch.device("cuda:0")
    nb, nf, nt, ca, cb = int(os.environ.get('NB', '64')), 256, 300, 256, 16
    w = np.random.RandomState(0).randn(128, ca + cb, 3, 3).astype(np.float32) * 0.05
    pk = ops.pack_conv3x3_bf16x(w, ca, cb, dev)
    xa = torch.randn((nb, nf, nt, ca), device=dev).bfloat16()
    xb = torch.randn((nb, nf, nt, cb), device=dev)
    for _ in range(2):
        ops.conv3x3_causal_bf16x(xa, xb, pk, 128, "relu")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv3x3_causal_bf16x(xa, xb, pk, 128, "relu")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("nb=%d abl=%s  %.3f ms  %.0f TFLOP/s" % (nb, os.environ.get("FNSSL_CONVX_ABL", "0"), ms, 2 * 9 * 272 * 128 * nb * nf * nt / ms / 1e9))
else:
    for abl, nb in ((63, 64), (63, 16), (63, 4), (0, 16), (0, 4)):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, FNSSL_CONVX_ABL=str(abl), NB=str(nb)), timeout=120)
