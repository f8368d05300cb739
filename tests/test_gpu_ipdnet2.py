"""GPU parity tests of the IPDnet2 row: every fnssl_sn_* entry point and the drop-in
OnlineSpatialNet against (i) tests/golden/g14_ipdnet2.npz — outputs of the REAL reference's pure-torch code
(LayerNorm, CausalConv1d, _fconv, _full, FreqInverse, layer / network orchestration) — and (ii) the numpy oracle.
The Mamba block is "parity unpinned" (no mamba_ssm anywhere): it is held to the oracle's restatement of the
published algorithm only.  Tolerance: fp32, rtol 1e-4 + atol 2e-5 (outputs are O(1))."""
import os

import numpy as np
import pytest

from conftest import assert_close, load_golden, rs_randn

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm device; none visible (the HIP path has no CPU fallback)")
    from fnssl import _lib
    _lib.load()
    return torch.device("cuda:0")


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def dropin():
    import importlib.util
    import os
    import sys
    if "fnssl_ipdnet2_dropin" in sys.modules:
        return sys.modules["fnssl_ipdnet2_dropin"]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fnssl_ipdnet2_dropin",
                                                  os.path.join(here, "fn-ssl_amd", "IPDnet2", "IPDnet2.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fnssl_ipdnet2_dropin"] = mod
    spec.loader.exec_module(mod)
    return mod


def build_net(dev, seed, **cfg):
    from fnssl import weights as W
    M = dropin()
    sd = W.make_ipdnet2_state(seed, **cfg)
    net = M.OnlineSpatialNet(dim_input=cfg.get("dim_input", 10), dim_output=16, num_layers=cfg.get("num_layers", 8),
                             dim_hidden=96, num_heads=4, kernel_size=(5, 3), conv_groups=(8, 8),
                             norms=["LN", "LN", "GN", "LN", "LN", "LN"], dim_squeeze=8,
                             num_freqs=cfg.get("num_freqs", 256), attention="mamba(16,4)", rope=False,
                             time_compression_layer=0, fre_compression_ratio=16, time_compression_ratio=5)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})     # strict: the reference's key names
    return sd, net.to(dev).eval()


def test_layernorm_wavefront_reduction_vs_reference(dev):
    g = load_golden("g14_ipdnet2")
    M = dropin()
    ln = M.LayerNorm(seq_last=True, normalized_shape=96).to(dev)
    with torch.no_grad():
        ln.weight.copy_(to_dev(g["ln_w"], dev)), ln.bias.copy_(to_dev(g["ln_b"], dev))
    got = ln(to_dev(rs_randn(3, (3, 96, 7)), dev))
    assert_close(got.cpu().numpy(), g["ln_out"], RTOL, ATOL, "LayerNorm(seq_last)")
    from fnssl import spatialnet as sn
    x = rs_randn(11, (5, 200))                                           # a width that is no multiple of 64
    w, b = rs_randn(12, (200,)), rs_randn(13, (200,))
    from oracle import ipdnet2_oracle as O2
    assert_close(sn.layernorm(to_dev(x, dev), to_dev(w, dev), to_dev(b, dev)).cpu().numpy(), O2.layer_norm(x, w, b),
                 RTOL, ATOL, "LayerNorm h=200")


def test_causal_conv1d_vs_reference_and_streaming(dev):
    g = load_golden("g14_ipdnet2")
    M = dropin()
    cc = M.CausalConv1d(in_channels=10, out_channels=96, kernel_size=5, look_ahead=0).to(dev)
    with torch.no_grad():
        cc.weight.copy_(to_dev(g["cc_w"], dev)), cc.bias.copy_(to_dev(g["cc_b"], dev))
    x = to_dev(rs_randn(6, (4, 10, 23)), dev)
    assert_close(cc(x).cpu().numpy(), g["cc_out"], RTOL, ATOL, "CausalConv1d")
    st = {}
    parts = [cc(x[..., :10].contiguous(), state=st), cc(x[..., 10:12].contiguous(), state=st),   # a 2-frame chunk:
             cc(x[..., 12:].contiguous(), state=st)]                                             # state mixes old + new
    assert_close(torch.cat(parts, -1).cpu().numpy(), g["cc_out"], RTOL, ATOL, "CausalConv1d chunked")


def test_fconv_full_pool_freqinverse_vs_reference(dev):
    g = load_golden("g14_ipdnet2")
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2100)
    l0, l1 = net.layers[0], net.layers[1]
    x0 = rs_randn(2101, (2, 32, 6, 96))
    assert_close(l0._fconv(l0.fconv1, to_dev(x0, dev)).cpu().numpy(), g["fconv1_out"], RTOL, ATOL, "_fconv")
    assert_close(l0._full(to_dev(rs_randn(2102, (1, 128, 3, 96)), dev)).cpu().numpy(), g["full128_out"], RTOL, ATOL,
                 "_full F=128")
    x16 = rs_randn(2103, (2, 16, 5, 96))
    assert_close(l1._full(to_dev(x16, dev)).cpu().numpy(), g["full16_out"], RTOL, ATOL, "_full F=16")
    assert_close(l1._fconv(l1.fconv2, to_dev(x16, dev)).cpu().numpy(), g["fconv2_l1_out"], RTOL, ATOL, "_fconv F=16")
    # residual + the fused frequency poolings of the first layer (reference: fre_compress_first / _second)
    from fnssl import spatialnet as sn
    w = l0._packed(dev)
    for pool, key in ((2, "pool2_out"), (8, "pool8_out")):
        got = sn.fconv(to_dev(x0, dev), w[0], residual=True, pool=pool).cpu().numpy()
        want = O2.avgpool_f(x0 + g["fconv1_out"], pool)
        assert_close(got, want, RTOL, ATOL, "fconv + pool %d" % pool)
        assert_close(O2.avgpool_f(x0, pool), g[key], 1e-6, 1e-6, "oracle pooling vs reference")
    # a strided (reference-layout [B, F, T, H] contiguous) input gives the same result as the native layout
    xs = to_dev(x0, dev)
    assert torch.equal(sn.fconv(xs, w[0]), sn.fconv(xs.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3), w[0]))
    assert_close(net.freq_inverse(to_dev(rs_randn(2105, (2, 96, 4, 16)), dev)).cpu().numpy(), g["finv_out"], RTOL, ATOL,
                 "FreqInverse")


def test_mamba_block_vs_oracle_and_streaming(dev):
    g = load_golden("g14_ipdnet2")
    from fnssl import spatialnet as sn
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2100)
    l1 = net.layers[1]
    w = l1._packed(dev)
    x = rs_randn(2108, (3, 17, 96))                                      # [S, T, H] = 3 sequences
    xs = to_dev(x, dev).unsqueeze(0)                                      # [B = 1, F = 3, T, H]
    want, _ = O2.mamba_block(sd, "layers.1.norm_mhsa", "layers.1.mhsa", x[None])
    got = sn.mamba(xs, w[3], residual=False)
    assert_close(got.cpu().numpy(), want, RTOL, ATOL, "LN + Mamba vs oracle")
    # the bare block (no LN) against the torch twin the fixtures were made with: feed LN^-1-free input by
    # comparing oracle's own bare block to the fixture (CPU) and the kernel to the oracle (above)
    y, _ = O2.mamba(sd, "layers.1.mhsa.", x)
    assert_close(y, g["mamba_out"], 1e-4, 2e-5, "oracle restatement vs its torch twin")
    # residual + time pooling (layer 0's tail): pool_T(x + branch)
    got = sn.mamba(xs, w[3], residual=True, time_pool=5).cpu().numpy()
    assert_close(got, O2.avgpool_t(x[None] + want, 5), RTOL, ATOL, "Mamba + residual + time pool")
    for tp in (2, 3):                                                    # the generic (run-time) pooling instantiation
        got = sn.mamba(xs, w[3], residual=True, time_pool=tp).cpu().numpy()
        assert_close(got, O2.avgpool_t(x[None] + want, tp), RTOL, ATOL, "Mamba + residual + time pool %d" % tp)
    # carried state: chunks of 4 + 1 + 12 frames == whole sequence
    st = sn.mamba_state(1, 3, dev)
    parts, t0 = [], 0
    for n in (4, 1, 12):
        parts.append(sn.mamba(xs[:, :, t0:t0 + n].contiguous(), w[3], residual=False, state=st, carry=t0 > 0))
        t0 += n
    assert_close(torch.cat(parts, 2).cpu().numpy(), want, RTOL, ATOL, "Mamba chunked")


def test_layers_and_network_vs_reference_orchestration(dev):
    g = load_golden("g14_ipdnet2")
    sd, net = build_net(dev, 2100)
    y, attn = net.layers[1](to_dev(rs_randn(2106, (1, 16, 10, 96)), dev))
    assert attn is None
    assert_close(y.cpu().numpy(), g["layer1_out"], RTOL, 5e-5, "SpatialNetLayer 1")
    y, _ = net.layers[0](to_dev(rs_randn(2107, (1, 256, 10, 96), 0.5), dev))
    assert tuple(y.shape) == (1, 16, 10, 96)
    assert_close(y.cpu().numpy(), g["layer0_out"], RTOL, 5e-5, "SpatialNetLayer 0 (F 256 -> 16)")
    x = to_dev(rs_randn(2110, (2, 10, 256, 20)), dev)
    out = net(x)
    assert tuple(out.shape) == (2, 4, 512, 4, 2)
    assert_close(out.cpu().numpy(), g["net_out"], RTOL, 5e-5, "OnlineSpatialNet (5-mic)")
    sd3, net3 = build_net(dev, 2200, dim_input=30, num_layers=3)
    assert_close(net3(to_dev(rs_randn(2210, (1, 30, 256, 15)), dev)).cpu().numpy(), g["net30_out"], RTOL, 5e-5,
                 "OnlineSpatialNet (15-mic input, BASELINE config 5 mapping)")


def test_network_streaming_and_batch_independence(dev):
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2300, num_layers=3)
    x = rs_randn(2301, (3, 10, 256, 40), 0.7)
    xd = to_dev(x, dev)
    whole = net(xd)
    assert_close(whole[:1].cpu().numpy(), O2.forward(sd, x[:1]), RTOL, 5e-5, "vs oracle")
    # utterances are independent (multi-GPU sharding = batch slicing): each alone == its row of the batch
    for b in range(3):
        assert torch.equal(net(xd[b:b + 1]), whole[b:b + 1])
    # online path: chunks of 10 / 5 / 25 frames with carried state == whole signal
    st, outs, t0 = None, [], 0
    for n in (10, 5, 25):
        o, st = net.forward_stream(xd[..., t0:t0 + n], st)
        outs.append(o)
        t0 += n
    assert_close(torch.cat(outs, 1).cpu().numpy(), whole.cpu().numpy(), 1e-5, 1e-5, "forward_stream")
    with pytest.raises(RuntimeError):
        net.forward_stream(xd[..., :7])
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 10, 256, 10))                                  # CPU tensor: no CPU path


def test_matrix_pipe_kernels_equal_scalar_kernels_at_multi_pass_size(dev, monkeypatch):
    """The fp32-MFMA formulation of the encoder, the frequency convs and the Mamba projections (persistent workgroups,
    several passes per workgroup at this size: 32 utterances x 250 frames) against the scalar-operand kernels that the
    other tests pin to the oracle and the reference fixtures (FNSSL_SN_SCALAR=1); same fp32 products, different
    summation order."""
    sd, net = build_net(dev, 2400, dim_input=30, num_layers=2)
    x = to_dev(rs_randn(2401, (32, 30, 256, 250), 0.7), dev)
    got = net(x)
    monkeypatch.setenv("FNSSL_SN_SCALAR", "1")
    want = net(x)
    monkeypatch.delenv("FNSSL_SN_SCALAR")
    assert tuple(got.shape) == (32, 50, 512, 4, 2)
    err = (got - want).abs().max().item()
    assert err <= 2e-5, err
    assert torch.equal(net(x), got)                                       # deterministic


@pytest.mark.parametrize("nb,nf,nt,layers", [(1, 256, 5, 2), (3, 128, 15, 2), (2, 256, 35, 3)])
def test_matrix_pipe_kernels_equal_scalar_kernels_at_ragged_sizes(dev, monkeypatch, nb, nf, nt, layers):
    """Point counts that are not multiples of the 16-point MFMA tiles, a 128-bin network, a single utterance; whole
    forward and a streamed forward (carried conv / scan / encoder state) against the scalar-operand kernels."""
    sd, net = build_net(dev, 2500 + nf + nt, dim_input=10, num_layers=layers, num_freqs=nf)
    x = to_dev(rs_randn(2501 + nt, (nb, 10, nf, nt), 0.7), dev)

    def run():
        whole = net(x)
        st, outs, t0 = None, [], 0
        for n in ([5] * (nt // 5)):
            o, st = net.forward_stream(x[..., t0:t0 + n], st)
            outs.append(o)
            t0 += n
        return whole, torch.cat(outs, 1)

    got, got_s = run()
    monkeypatch.setenv("FNSSL_SN_SCALAR", "1")
    want, want_s = run()
    monkeypatch.delenv("FNSSL_SN_SCALAR")
    assert got.shape == want.shape == (nb, nt // 5, 2 * nf, 4, 2)
    assert (got - want).abs().max().item() <= 2e-5
    assert (got_s - want_s).abs().max().item() <= 2e-5
    assert (got_s - got).abs().max().item() <= 2e-5


# ------------------------------------------------------------------------------------------------------------ #
# FNSSL_PRECISION_BF16 (BASELINE config 5 as written): bf16 MFMA operands in the encoder, the grouped frequency conv,
# the full-band branch and the Mamba in / x / out projections; fp32 accumulation, fp32 tensors, everything else fp32
# (LayerNorm, biases / activations, depthwise conv, dt_proj, scan, FreqInverse, decoder).  No reference exists
# for it: the kernels are held (i) to the oracle's restatement of exactly this rounding
# (ipdnet2_oracle.bf16_products) at BF_RTOL / BF_ATOL per op — differences are operands that sit within one fp32
# rounding of a bf16 tie and fall to the other side (measured: 1e-6 typical, 8e-4 where one flips; outputs O(1)) — and
# (ii) to the fp32 oracle at a looser tolerance.  Through a whole network a flipped operand is amplified by the layers
# that follow (measured against the restatement: 2.8e-3 max / 4.5e-4 rms at 3 layers, 6.2e-3 / 9e-4 at 8; against fp32:
# 6.3e-3 / 1.5e-3 and 1.2e-2 / 2.2e-3 on outputs of rms 0.3 - 0.45), hence NET_* below.
# ------------------------------------------------------------------------------------------------------------ #
BF_RTOL, BF_ATOL = 4e-3, 2e-3
LOOSE_RTOL, LOOSE_ATOL = 2e-2, 4e-3
NET_BF_RTOL, NET_BF_ATOL = 1e-2, 6e-3
NET_LOOSE_RTOL, NET_LOOSE_ATOL = 2e-2, 1.5e-2


def _bf16_sd(sd):
    from oracle import ipdnet2_oracle as O2
    return {k: O2.bf16_round(v) for k, v in sd.items()}


def test_bf16_ops_vs_oracle_restating_the_rounding(dev):
    from fnssl import spatialnet as sn
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2500, dim_input=30, num_layers=2)
    # encoder (K = 150 -> 5 bf16 k-steps) and the small-K instantiation (dim_input 10: K = 50 -> 3 k-steps)
    for cin, seed in ((30, 2501), (10, 2502)):
        w, b = rs_randn(seed, (96, cin, 5), 0.1), rs_randn(seed + 10, (96,), 0.1)
        x = rs_randn(seed + 20, (2, cin, 16, 23))
        wT = to_dev(w, dev).permute(1, 2, 0).contiguous()
        got = sn.encoder(to_dev(x, dev), wT, to_dev(b, dev), precision=sn.BF16).cpu().numpy()     # [B, F, T, 96]
        with O2.bf16_products():
            want = np.stack([O2.causal_conv1d(x[:, :, f, :], w, b)[0] for f in range(16)], 1).transpose(0, 1, 3, 2)
        exact = np.stack([O2.causal_conv1d(x[:, :, f, :], w, b)[0] for f in range(16)], 1).transpose(0, 1, 3, 2)
        assert_close(got, want, BF_RTOL, BF_ATOL, "encoder bf16 (cin %d)" % cin)
        assert np.abs(got - exact).max() > 1e-5, "bf16 mode did not change the product"
        # carried frames: chunked == whole, bit for bit (the same products per point)
        st = torch.empty((2, cin, 16, 4), dtype=torch.float32, device=dev)
        xd = to_dev(x, dev)
        a = sn.encoder(xd[..., :9].contiguous(), wT, to_dev(b, dev), None, st, precision=sn.BF16)
        st2 = torch.empty_like(st)
        c = sn.encoder(xd[..., 9:].contiguous(), wT, to_dev(b, dev), st, st2, precision=sn.BF16)
        assert torch.equal(torch.cat([a, c], 2), torch.from_numpy(got).to(dev))
    # grouped frequency conv at 256 / 128 (+ pool) / 16 bins
    l0 = net.layers[0]
    w0 = l0._packed(dev)
    for nf, pool, seed in ((256, 2, 2510), (128, 8, 2511), (16, 1, 2512), (32, 1, 2513)):
        x = rs_randn(seed, (2, nf, 5, 96))
        got = sn.fconv(to_dev(x, dev), w0[0], residual=True, pool=pool, precision=sn.BF16).cpu().numpy()
        with O2.bf16_products():
            want = O2.avgpool_f(x + O2.fconv(sd, "layers.0.fconv1", x), pool)
        assert_close(got, want, BF_RTOL, BF_ATOL, "fconv bf16 nf %d pool %d" % (nf, pool))
    # full-band branch (squeeze, Linear over F, unsqueeze on bf16 operands) at the three sizes of the matrix-pipe kernel
    for nf, lp, seed in ((128, "layers.0.", 2514), (16, "layers.1.", 2515)):
        x = rs_randn(seed, (2, nf, 6, 96))
        wl = (net.layers[0] if nf == 128 else net.layers[1])._packed(dev)
        got = sn.full(to_dev(x, dev), wl[1], residual=True, precision=sn.BF16).cpu().numpy()
        with O2.bf16_products():
            want = x + O2.full(sd, lp, x)
        assert_close(got, want, BF_RTOL, BF_ATOL, "full bf16 nf %d" % nf)
        assert np.abs(got - (x + O2.full(sd, lp, x))).max() > 1e-6, "bf16 mode did not change the product"
    # Mamba block: LN + in_proj / x_proj / out_proj on bf16 operands, scan fp32; + residual + time pooling; streaming
    x = rs_randn(2520, (3, 35, 96))
    xs = to_dev(x, dev).unsqueeze(0)
    with O2.bf16_products():
        want, _ = O2.mamba_block(sd, "layers.0.norm_mhsa", "layers.0.mhsa", x[None])
    exact, _ = O2.mamba_block(sd, "layers.0.norm_mhsa", "layers.0.mhsa", x[None])
    got = sn.mamba(xs, w0[3], residual=False, precision=sn.BF16)
    assert_close(got.cpu().numpy(), want, BF_RTOL, BF_ATOL, "LN + Mamba bf16 vs oracle (bf16 products)")
    assert_close(got.cpu().numpy(), exact, LOOSE_RTOL, LOOSE_ATOL, "LN + Mamba bf16 vs fp32 oracle")
    got5 = sn.mamba(xs, w0[3], residual=True, time_pool=5, precision=sn.BF16).cpu().numpy()
    assert_close(got5, O2.avgpool_t(x[None] + want, 5), BF_RTOL, BF_ATOL, "Mamba bf16 + residual + time pool")
    st = sn.mamba_state(1, 3, dev)
    parts, t0 = [], 0
    for n in (4, 1, 30):
        parts.append(sn.mamba(xs[:, :, t0:t0 + n].contiguous(), w0[3], residual=False, state=st, carry=t0 > 0,
                              precision=sn.BF16))
        t0 += n
    assert_close(torch.cat(parts, 2).cpu().numpy(), got.cpu().numpy(), 1e-5, 1e-5, "Mamba bf16 chunked == whole")
    with pytest.raises(RuntimeError):
        sn.mamba(xs, w0[3], precision=7)


def test_bf16_network_after_bfloat16_vs_oracles_and_streaming(dev):
    """``net.bfloat16()`` (parameters rounded to bf16, as torch does) selects FNSSL_PRECISION_BF16 end to end."""
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2600, dim_input=30, num_layers=3)
    x = rs_randn(2601, (2, 30, 256, 40), 0.7)
    xd = to_dev(x, dev)
    ref32 = net(xd)
    net = net.bfloat16()
    out = net(xd)
    assert out.dtype == torch.float32 and tuple(out.shape) == (2, 8, 512, 4, 2)
    sdb = _bf16_sd(sd)
    with O2.bf16_products():
        want = O2.forward(sdb, x[:1])
    assert_close(out[:1].cpu().numpy(), want, NET_BF_RTOL, NET_BF_ATOL, "bf16 network vs oracle (bf16 parameters + products)")
    e = np.abs(out[:1].cpu().numpy() - want)
    assert np.sqrt((e ** 2).mean()) < 1.5e-3, "rms deviation from the restated rounding"
    assert_close(out[:1].cpu().numpy(), O2.forward(sd, x[:1]), NET_LOOSE_RTOL, NET_LOOSE_ATOL, "bf16 network vs the fp32 oracle")
    dev32 = (out - ref32).abs().max().item()
    assert 1e-6 < dev32 < 2e-2, dev32                                     # a different arithmetic, and a close one
    for b in range(2):
        assert torch.equal(net(xd[b:b + 1]), out[b:b + 1])                # utterances stay independent
    st, outs, t0 = None, [], 0
    for n in (10, 5, 25):
        o, st = net.forward_stream(xd[..., t0:t0 + n], st)
        outs.append(o)
        t0 += n
    assert_close(torch.cat(outs, 1).cpu().numpy(), out.cpu().numpy(), 1e-5, 1e-5, "bf16 forward_stream")
    assert torch.equal(net(xd), out)                                      # deterministic


def test_bf16_config5_full_batch_independence_and_closeness_to_fp32(dev):
    """BASELINE config 5 at its real size (64 utterances x 30 channels x 256 bins x 250 frames, 8 layers) in the bf16
    mode: every kernel runs several passes per workgroup here.  Size-independent properties: sampled utterances of the
    batch equal the same utterance run alone bit for bit (utterance shards = the multi-GPU split), the forward is
    deterministic, and the outputs stay close to the fp32 kernels (which the other tests pin to the oracle and the
    reference fixtures): rms, maximum and the fraction outside the element-wise bf16 tolerance are bounded."""
    if torch.cuda.mem_get_info()[0] < 40 << 30:
        pytest.skip("needs 40 GB of free HBM")
    sd, net = build_net(dev, 2700, dim_input=30, num_layers=8)
    g = torch.Generator(device=dev)
    g.manual_seed(2701)
    x = torch.randn((64, 30, 256, 250), generator=g, device=dev)
    ref = net(x)
    netb = net.bfloat16()
    out = netb(x)
    assert tuple(out.shape) == (64, 50, 512, 4, 2) and bool(torch.isfinite(out).all())
    assert torch.equal(netb(x), out)
    for b in (0, 31, 63):
        assert torch.equal(netb(x[b:b + 1]), out[b:b + 1]), "utterance %d depends on its batch" % b
    # 26 M outputs: the tail of the deviation is ~10 sigma (measured: rms 2.2e-3, max 2.2e-2 on outputs of rms 0.45), so
    # the element-wise tolerance of the small tests is stated here as a quantile plus hard bounds on rms and maximum
    err = (out - ref).abs()
    tol = NET_LOOSE_ATOL + NET_LOOSE_RTOL * ref.abs()
    outside = (err > tol).float().mean().item()
    assert outside < 1e-5, "fraction outside rtol %g / atol %g: %.2e" % (NET_LOOSE_RTOL, NET_LOOSE_ATOL, outside)
    assert err.max().item() < 5e-2, "max abs deviation from the fp32 kernels %.3e" % err.max().item()
    assert err.pow(2).mean().sqrt().item() < 4e-3


def test_config5_full_waveform_batch_sampled_utterance_vs_cpu_reference(dev):
    """BASELINE config 5 as benchmarked — 64 utterances x 15 microphones from the WAVEFORM, bf16 — with one sampled
    utterance of that batch held to the fp32 PyTorch-CPU restatement (oracle/torch_ref.py::ipdnet2_forward on
    array_preprocess: torch.stft, conv1d, layer_norm, the Mamba block as the published algorithm).  The network is causal and
    so is the recursive normalisation, so a frame prefix is an exact sub-problem up to the frames whose centred STFT window
    reaches the reflected end of the shorter signal: the CPU side runs 55 frames, the first 45 (9 output frames) are compared.
    Tolerance = bench.py's config-5 parity gate: rtol 2e-2 / atol 1.3e-2 (observed 1.2e-2 max, 2.2e-3 rms on outputs of rms
    0.45: bf16 operand rounding through 8 layers; the fp32 kernels are at 9e-7)."""
    from fnssl import ops
    from oracle import torch_ref as R
    if torch.cuda.mem_get_info()[0] < 40 << 30:
        pytest.skip("needs 40 GB of free HBM")
    sd, net = build_net(dev, 2900, dim_input=30, num_layers=8)
    net = net.bfloat16()
    g = torch.Generator(device=dev)
    g.manual_seed(2901)
    nt = 250
    sig = torch.randn((64, 320 * (nt - 1), 15), generator=g, device=dev) * 0.1
    out = net(ops.preprocess_ipdnet2(sig))
    assert tuple(out.shape) == (64, nt // 5, 512, 4, 2) and bool(torch.isfinite(out).all())
    u, frames, keep = 37, 55, 9
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    want = R.ipdnet2_forward(sd, R.array_preprocess(sig[u:u + 1, :320 * (frames - 1)].cpu(), 249, 320, True))
    got = out[u:u + 1, :keep].float().cpu()
    want = want[:, :keep]
    err = (got - want).abs()
    assert bool((err <= 1.3e-2 + 2e-2 * want.abs()).all()), "utterance %d of the full batch vs the CPU reference: max abs err %.3e" % (u, err.max().item())
    assert err.pow(2).mean().sqrt().item() < 4e-3


@pytest.mark.parametrize("nb,nf,nt,layers", [(1, 256, 5, 2), (3, 128, 15, 2), (2, 256, 35, 3)])
def test_bf16_network_at_ragged_sizes_and_128_bins(dev, nb, nf, nt, layers):
    """The bf16 kernels at point counts that are not multiples of the 16-point tiles (and of the 13-output tiles of the
    fused conv + x_proj pass), a 128-bin network (full-band kernel at 64 bins) and a single short utterance: against the
    oracle's restatement of the rounding, and streamed (carried encoder / conv / scan state: the two-kernel conv path)
    against whole."""
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2800 + nf + nt, dim_input=10, num_layers=layers, num_freqs=nf)
    net = net.bfloat16()
    x = rs_randn(2801 + nt, (nb, 10, nf, nt), 0.7)
    xd = to_dev(x, dev)
    whole = net(xd)
    assert tuple(whole.shape) == (nb, nt // 5, 2 * nf, 4, 2)
    with O2.bf16_products():
        want = O2.forward(_bf16_sd(sd), x, fre_compression_ratio=16)
    assert_close(whole.cpu().numpy(), want, NET_BF_RTOL, NET_BF_ATOL, "bf16 network nf %d nt %d" % (nf, nt))
    st, outs, t0 = None, [], 0
    for n in ([5] * (nt // 5)):
        o, st = net.forward_stream(xd[..., t0:t0 + n], st)
        outs.append(o)
        t0 += n
    assert (torch.cat(outs, 1) - whole).abs().max().item() <= 2e-5


def test_ipdnet2_waveform_frontend_vs_reference_golden(dev):
    """IPDnet2's front end (IPDnet2/Module.py:47-64 centred hop-320 STFT, run_IPDnet2.py:277-288 all-channel
    forgetting_norm(249) + real/imag + DC drop) on the device against G15 = the reference's own modules, and against the
    oracle; the per-stage STFT drop-in; waveform -> network == features -> network."""
    import importlib.util
    import os
    from fnssl import ops
    from oracle import fnssl_oracle as O
    g = load_golden("g15_ipdnet2_frontend")
    ci = 0
    while "c%d_cfg" % ci in g.files:
        seed, nb, ns, nch, sl = (int(v) for v in g["c%d_cfg" % ci])
        sig = rs_randn(seed, (nb, ns, nch), 0.1)
        want = g["c%d_feat" % ci]
        x = ops.preprocess_ipdnet2(to_dev(sig, dev), sample_length=sl)
        assert tuple(x.shape) == want.shape
        scale = np.abs(want).max()
        assert np.abs(x.cpu().numpy() - want).max() <= 5e-6 * scale, "case %d vs reference golden" % ci
        assert np.abs(x.cpu().numpy() - O.array_preprocess(sig, sample_length=sl, hop=320, center=True)).max() <= 5e-6 * scale
        # the dataloader's [nb, nch, ns] batch read in place through strides
        x2 = ops.preprocess_ipdnet2(to_dev(np.ascontiguousarray(sig.transpose(0, 2, 1)), dev).permute(0, 2, 1), sample_length=sl)
        assert torch.equal(x2, x)
        assert x.is_contiguous()
        assert torch.equal(ops.preprocess_ipdnet2(to_dev(sig, dev), sample_length=sl, frame_major=True), x)
        # the one-call front end that never writes the spectrum: the same numbers
        xf = ops.array_frontend(to_dev(sig, dev), sample_length=sl, hop=320, center=True)
        assert torch.equal(xf.permute(0, 3, 2, 1), x)
        ci += 1
    assert ci == 4
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fnssl_ipdnet2_module", os.path.join(here, "fn-ssl_amd", "IPDnet2", "Module.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seed, nb, ns, nch, sl = (int(v) for v in g["c0_cfg"])
    st = mod.STFT(win_len=512, win_shift_ratio=0.625, nfft=512)(to_dev(rs_randn(seed, (nb, ns, nch), 0.1), dev))
    assert st.dtype == torch.complex64 and tuple(st.shape) == g["c0_stft"].shape
    assert np.abs(st.cpu().numpy() - g["c0_stft"]).max() <= 5e-6 * np.abs(g["c0_stft"]).max()
    assert ops.num_frames(ns, 320, True) == ns // 320 + 1 and ops.num_frames(256, 320, True) == 0
    with pytest.raises(RuntimeError):
        ops.stft(to_dev(rs_randn(1, (1, 200, 2)), dev), hop=320, center=True)      # shorter than the reflect padding
    # waveform -> DP-IPD through the drop-in == features -> DP-IPD, and == the oracle chain
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2400, num_layers=2)
    sig = rs_randn(2401, (2, 320 * 19, 5), 0.1)
    feats = ops.preprocess_ipdnet2(to_dev(sig, dev))
    assert tuple(feats.shape) == (2, 10, 256, 20)
    out = net(feats)
    assert torch.equal(out, net(feats.contiguous()))
    want = O2.forward(sd, O.array_preprocess(sig[:1], sample_length=249, hop=320, center=True))
    assert_close(out[:1].cpu().numpy(), want, RTOL, 5e-5, "waveform -> OnlineSpatialNet vs oracle chain")


def test_inference_flag_steps_the_mamba_blocks_frame_by_frame(dev):
    """``OnlineSpatialNet(x, inference=True)`` / ``SpatialNetLayer(x, inference=True)`` / ``_mamba(..., inference=True)``:
    the reference's per-frame branch (IPDnet2.py:170-177) restated in the oracle as an explicit-state recurrence
    (``mamba_step``), against the device path that drives ``fnssl_sn_mamba`` one frame at a time."""
    from fnssl import spatialnet as sn
    from oracle import ipdnet2_oracle as O2
    sd, net = build_net(dev, 2600, num_layers=2)
    x = rs_randn(2601, (2, 10, 256, 10), 0.8)
    xd = to_dev(x, dev)
    got = net(xd, inference=True)
    assert tuple(got.shape) == (2, 2, 512, 4, 2)
    assert_close(got[:1].cpu().numpy(), O2.forward(sd, x[:1], inference=True), RTOL, 5e-5, "network, inference=True vs stepwise oracle")
    assert_close(got.cpu().numpy(), net(xd).cpu().numpy(), 1e-4, 2e-5, "inference=True == parallel mode")
    # one layer and one bare block, frame by frame, against the per-step restatement
    l1 = net.layers[1]
    xl = rs_randn(2602, (1, 16, 7, 96))
    y, _ = l1(to_dev(xl, dev), inference=True)
    want, _ = O2.layer_forward(sd, "layers.1.", xl, False, inference=True)
    assert_close(y.cpu().numpy(), want, RTOL, 5e-5, "SpatialNetLayer(inference=True)")
    yb = l1._mamba(to_dev(xl, dev), l1.mhsa, l1.norm_mhsa, l1.dropout_mhsa, inference=True)
    wb, _ = O2.mamba_block(sd, "layers.1.norm_mhsa", "layers.1.mhsa", xl, stepwise=True)
    assert_close(yb.cpu().numpy(), wb, RTOL, ATOL, "_mamba(inference=True) vs mamba_step recurrence")
    # T = 1 drive of the kernel against the explicit-state oracle step, state compared after every frame
    w = l1._packed(dev)
    S, T = 5, 6
    xs = rs_randn(2603, (S, T, 96))
    st = sn.mamba_state(1, S, dev)
    cs, ss = np.zeros((S, 192, 4), np.float32), np.zeros((S, 192, 16), np.float32)
    for t in range(T):
        xin = to_dev(xs[None, :, t:t + 1], dev)
        o = sn.mamba(xin, w[3], residual=False, state=st, carry=t > 0)
        ln = O2.layer_norm(xs[:, t], sd["layers.1.norm_mhsa.weight"], sd["layers.1.norm_mhsa.bias"])
        wo, cs, ss = O2.mamba_step(sd, "layers.1.mhsa.", ln, cs, ss)
        assert_close(o[0, :, 0].cpu().numpy(), wo, RTOL, ATOL, "step %d output" % t)
        assert_close(st[1].cpu().numpy(), ss, RTOL, ATOL, "step %d SSM state" % t)
        assert_close(st[0].cpu().numpy(), np.transpose(cs[:, :, 1:], (0, 2, 1)), 1e-6, 1e-6, "step %d conv taps" % t)
