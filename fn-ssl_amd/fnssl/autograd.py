"""``torch.autograd.Function``s over the C-ABI training kernels: the drop-in ``Model.FN_SSL.forward`` /
``Model.FNblock.forward`` in ``train()`` mode return tensors that carry a ``grad_fn``, so the reference's own training
code runs unchanged on the HIP path:

    pred = self(in_batch); loss = self.cal_loss(pred, gt); loss.backward(); optimizer.step()
        FN-SSL/Lightning/main.py:149-157 (Lightning automatic optimisation), FN-SSL/Learner.py:104-115
    DDP (main.py:286-288): DistributedDataParallel's reducer hooks fire on the parameters' ``.grad`` accumulation exactly
    as with nn.LSTM — the RCCL bucket all-reduce is DDP's own, nothing bespoke

Forward = the reserve-saving LSTM kernels (``fnssl_lstm_forward`` with ``reserve``) + ``fnssl_train_combine`` (dropout +
residual adds) + ``fnssl_head``; backward = ``fnssl_head_backward``, ``fnssl_lstm_backward`` (BPTT),
``fnssl_lstm_weight_grads``, ``fnssl_train_combine`` again — the same ``train.TrainGraph`` halves the fused
``TrainEngine`` runs, so both routes produce the same numbers.  No ATen kernel computes anything here besides the
concatenation of the parameters into the flat vector the weight packers gather from.

Dropout (Model.py:40,48): nn.Dropout's Bernoulli stream cannot be reproduced outside torch's RNG; the keep mask is the
library's hash of (seed, call, layer, GLOBAL element index) — ``module.dropout_seed`` (default ``torch.initial_seed()``)
and ``module.dropout_calls`` (incremented per train-mode forward) form the base seed with the engine's formula,
``module.pair_offset`` (default rank * pairs under an initialised process group) keys the element index so that N ranks
draw the masks of one rank on the concatenated batch.  REQUIREMENT under a process group (not checked: it would cost a
collective per forward): every rank passes the SAME number of pairs per call and the same ``dropout_seed`` — the default seed is
``torch.initial_seed()``, which differs per rank unless the ranks were seeded identically (Lightning's ``seed_everything`` /
``torch.manual_seed(s)`` on every rank).  With uneven last batches, or a module that is not data-parallel although a group
is initialised, set ``module.pair_offset`` (and ``module.dropout_seed``) explicitly; masks of different ranks otherwise overlap
or differ from the one-process batch.  ``_lib.tuning(...)`` around a train-mode forward is PROCESS-wide (``fnssl_tuning_set``) and
does not cover the backward, which autograd runs later: pass knobs for the backward through ``TrainGraph(bwd_tuning=...)``.

The gradient w.r.t. the input FEATURES (``in_batch.requires_grad_()``, Learner.py:102 — set, never read by the
reference) is not produced: block 1's layers have no input-gradient path in the BPTT kernels (c0g = 0).  ``backward``
returns ``None`` for it, i.e. ``in_batch.grad`` stays ``None``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, train

_MASK32 = 0xFFFFFFFF


def base_seed(seed: int, call: int) -> int:
    """The base dropout seed of the ``call``-th train-mode forward (TrainEngine.step uses the same formula)."""
    return (int(seed) * 1000003 + int(call) * 8191) & _MASK32


def _next_base(module) -> int:
    """Advance ``module.dropout_calls`` and return the base seed of this train-mode forward.  ``module.force_dropout_base``
    (tests / parity legs) pins it: the next forwards draw exactly the masks of that base seed."""
    module.dropout_calls = int(getattr(module, "dropout_calls", 0)) + 1
    forced = getattr(module, "force_dropout_base", None)
    if forced is not None:
        return int(forced) & _MASK32
    seed = getattr(module, "dropout_seed", None)
    return base_seed(torch.initial_seed() if seed is None else seed, module.dropout_calls)


def _cc(t):
    """fp32, channel-contiguous (an upstream gradient may arrive in any layout)."""
    t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


def _fresh(_key, shape):
    return torch.empty(shape, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))


def _rank_offset(nbp: int) -> int:
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank() * int(nbp)
    return 0


class NetPlan:
    """Per-(model, device) constants of the autograd route: layer table, flat layout, gather maps."""

    def __init__(self, named_shapes, online: bool, dev):
        self.names = [k for k, _ in named_shapes]
        self.offset, self.total = train.flat_layout(named_shapes)
        self.layers = train.build_layers(online)
        self.maps = train.build_index_maps(self.layers, self.offset, dev)
        self.dev = dev

    def flatten(self, params):
        """[0 | p_0 | p_1 | ...] — the vector the index maps gather the packed weight streams from."""
        z = torch.zeros(1, dtype=torch.float32, device=self.dev)
        return torch.cat([z] + [p.detach().reshape(-1) for p in params])

    def view(self, flat, name):
        off, shape = self.offset[name]
        return flat[off:off + int(np.prod(shape))].view(shape)


class FNSSLTrainFunction(torch.autograd.Function):
    """pred = FN_SSL(x) in train mode; backward accumulates every parameter gradient with the HIP BPTT kernels."""

    @staticmethod
    def forward(ctx, x, plan, seeds, b0, *params):
        with torch.cuda.device(x.device):
            theta = plan.flatten(params)
            fw, bw = train.pack_streams(plan.layers, plan.maps, theta)
            graph = train.TrainGraph(plan.layers, _fresh)
            pred, saved = graph.forward(x.detach().contiguous(), fw, seeds, b0, lambda n: plan.view(theta, n))
            # head_backward needs tanh's output: keep an ALIAS (same storage, another tensor object), not the returned tensor
            # itself — the output object on ctx would close the cycle output -> grad_fn -> ctx -> output and keep ~100 GB of
            # activations alive until the cyclic collector runs whenever backward is never called (round-5 advisor note).
            # The caller must not modify `pred` in place before backward (no version counter watches the alias).
            saved["pred"] = pred.detach()
        ctx.plan, ctx.graph, ctx.saved, ctx.bw, ctx.theta = plan, graph, saved, bw, theta
        ctx.seeds, ctx.b0 = seeds, b0
        ctx.set_materialize_grads(False)
        return pred

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dpred):
        plan = ctx.plan
        if ctx.saved is None:
            raise RuntimeError("fnssl.autograd: backward through the same forward twice (the saved activations were "
                               "released; call forward again)")
        nparam = len(plan.names)
        if dpred is None:
            return (None,) * (4 + nparam)
        with torch.cuda.device(dpred.device):
            grad = torch.zeros(plan.total, dtype=torch.float32, device=dpred.device)
            ctx.graph.backward(ctx.saved, dpred.contiguous().float(), ctx.bw, ctx.seeds, ctx.b0,
                               lambda n: plan.view(ctx.theta, n), lambda n: plan.view(grad, n))
        ctx.saved = ctx.bw = ctx.theta = None                 # ≈ 100 GB at config 4's shard: release now, not at ctx's death
        grads = tuple(plan.view(grad, n) if ctx.needs_input_grad[4 + i] else None for i, n in enumerate(plan.names))
        return (None, None, None, None) + grads


def fnssl_train_forward(model, x):
    """``FN_SSL.forward`` in train mode: x [nb', 4, nf, nt] -> DP-IPD [nb', nt//12, 2nf] with a ``grad_fn``."""
    if getattr(model, "is_doa", False):
        raise RuntimeError("FN_SSL.forward (train mode): the DOA-classification variant is not part of the training path")
    if next(model.parameters()).dtype != torch.float32:
        raise RuntimeError("FN_SSL.forward (train mode): fp32 parameters only (the bf16 mode is inference)")
    ops._need_dev(x)
    if x.ndim != 4 or x.shape[1] != 4 or x.shape[3] < ops.SEG_FRAMES:
        raise RuntimeError("FN_SSL.forward (train mode): expected [nb, 4, nf, nt >= 12], got %s" % (tuple(x.shape),))
    named = list(model.named_parameters())
    key = (tuple((k, tuple(p.shape)) for k, p in named), str(x.device))
    plan = getattr(model, "_autograd_plan", None)
    if plan is None or model._autograd_plan_key != key:
        plan = NetPlan([(k, p.shape) for k, p in named], bool(model.is_online), x.device)
        model._autograd_plan, model._autograd_plan_key = plan, key
    base = _next_base(model)
    seeds = [train.layer_seed(base, l) for l in range(6)]
    b0 = getattr(model, "pair_offset", None)
    b0 = _rank_offset(x.shape[0]) if b0 is None else int(b0)
    model.last_dropout_base = base
    return FNSSLTrainFunction.apply(x, plan, seeds, b0, *[p for _, p in named])


# --------------------------------------------------------------------------- #
# one FN block (Model.py:31-50) — the reference's FNblock.forward used on its own
# --------------------------------------------------------------------------- #
class BlockPlan:
    def __init__(self, blk, dev):
        named = list(blk.named_parameters())
        self.names = [k for k, _ in named]
        self.offset, self.total = train.flat_layout([(k, p.shape) for k, p in named])
        fh, nh = blk.full_hidden_size, blk.narr_hidden_size
        if fh != train.H_FULL or nh not in (train.H_FULL, train.H_NARR_ONLINE) or \
                blk.input_size != (4 if blk.is_first else train.CH):
            raise RuntimeError("FNblock.forward (train mode): the BPTT kernels are built for hidden_size 256 "
                               "(full-band H = 128, narrow-band H = 256 / 128) and input_size 4 (first) / 256")
        first = blk.is_first
        nd = 1 if blk.is_online else 2
        self.lf = train._Layer("fullLstm", "full", fh, 2, 4 if first else train.CH, 0, 0 if first else train.CH)
        self.ln = train._Layer("narrLstm", "narrow", nh, nd, train.CH, 4 if first else 0, train.CH)
        self.layers = [self.lf, self.ln]
        self.maps = train.build_index_maps(self.layers, self.offset, dev)
        self.first, self.dev = first, dev

    flatten = NetPlan.flatten
    view = NetPlan.view


class FNblockTrainFunction(torch.autograd.Function):
    """(x_out, fb_skip, nb_skip) = FNblock(x, fb_skip_prev) in train mode (dropout_full / dropout_narr active)."""

    @staticmethod
    def forward(ctx, x, fb_prev, plan, seeds, b0, *params):
        CH = train.CH
        lf, ln = plan.lf, plan.ln
        with torch.cuda.device(x.device):
            theta = plan.flatten(params)
            fw, bw = train.pack_streams(plan.layers, plan.maps, theta)
            g = train.TrainGraph([lf, ln] * 3, _fresh)
            x = _cc(x.detach())
            nb, nt, nf, _ = x.shape
            F = g._natural("F", lf, nb, nt, nf, 2 * lf.hidden)
            rf = _fresh("R", (ops.lstm_reserve_floats(nb * nt, lf.hidden, 2, nf),))
            if plan.first:
                U = x.contiguous()
                XN = U.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
            else:
                U = g._natural("U", lf, nb, nt, nf, CH)
                train.combine(U, plain=(x, _cc(fb_prev.detach()).reshape(nb, nt, nf, CH)))          # x + fb_skip :36-37
                XN = None
            ops.lstm_layer("full", U, None, None, fw[lf.name], lf.hidden, F, reserve=rf)
            V = g._natural("V", ln, nb, nt, nf, CH)
            if plan.first:
                train.combine(V, masked=(F,), seed32=seeds[0], b0=b0)                          # dropout_full :40
            else:
                train.combine(V, masked=(F,), plain=(x,), seed32=seeds[0], b0=b0)              # + nb_skip :44-45
            nh = ln.ndir * ln.hidden
            N = g._natural("N", ln, nb, nt, nf, nh)
            rn = _fresh("R", (ops.lstm_reserve_floats(nb * nf, ln.hidden, ln.ndir, nt),))
            ops.lstm_layer("narrow", V, None, XN, fw[ln.name], ln.hidden, N, reserve=rn)
            out = g._natural("X", ln, nb, nt, nf, nh)
            train.combine(out, masked=(N,), seed32=seeds[1], b0=b0)                            # dropout_narr :48
        ctx.plan, ctx.graph, ctx.bw, ctx.seeds, ctx.b0 = plan, g, bw, seeds, b0
        ctx.saved = (U, XN, F, V, N, rf, rn)
        ctx.set_materialize_grads(False)
        fb_out = F.view(nb * nt, nf, 2 * lf.hidden)
        nb_out = N.permute(0, 2, 1, 3).reshape(nb * nf, nt, nh)
        return out, fb_out, nb_out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out, g_fb, g_nb):
        CH = train.CH
        plan, g, bw, seeds, b0 = ctx.plan, ctx.graph, ctx.bw, ctx.seeds, ctx.b0
        lf, ln = plan.lf, plan.ln
        if ctx.saved is None:
            raise RuntimeError("fnssl.autograd: backward through the same forward twice")
        U, XN, F, V, N, rf, rn = ctx.saved
        nb, nt, nf = F.shape[:3]
        nh = ln.ndir * ln.hidden
        dev = F.device
        nparam = len(plan.names)
        if g_out is None and g_fb is None and g_nb is None:
            return (None,) * (5 + nparam)
        with torch.cuda.device(dev):
            grad = torch.zeros(plan.total, dtype=torch.float32, device=dev)
            gview = lambda n: plan.view(grad, n)   # noqa: E731
            masked = (_cc(g_out),) if g_out is not None else ()
            plain = (_cc(g_nb).reshape(nb, nf, nt, nh).permute(0, 2, 1, 3),) if g_nb is not None else ()
            DN = g._natural("DN", ln, nb, nt, nf, nh)
            if masked or plain:
                train.combine(DN, masked=masked, plain=plain, seed32=seeds[1], b0=b0)          # dropout_narr backward
            else:
                DN.zero_()
            dA = g._natural("dA", ln, nb, nt, nf, ln.ndir * 4 * ln.hidden)
            DV = g._natural("DV", ln, nb, nt, nf, ln.ndir * CH)
            ops.lstm_backward("narrow", rn, DN, dA, DV, bw[ln.name], ln.hidden, CH)
            g._weight_grads(ln, dA, V, XN, N, gview, lambda p: None)
            dv = tuple(DV[..., d * CH:(d + 1) * CH] for d in range(ln.ndir))
            DF = g._natural("DF", lf, nb, nt, nf, 2 * lf.hidden)
            plain = (_cc(g_fb).reshape(nb, nt, nf, 2 * lf.hidden),) if g_fb is not None else ()
            train.combine(DF, masked=dv, plain=plain, seed32=seeds[0], b0=b0)                  # dropout_full backward
            dA = g._natural("dA", lf, nb, nt, nf, 2 * 4 * lf.hidden)
            dx = dfb = None
            if plan.first:
                ops.lstm_backward("full", rf, DF, dA, None, bw[lf.name], lf.hidden, 0)
                g._weight_grads(lf, dA, U, None, F, gview, lambda p: None)
            else:
                DU = g._natural("DU", lf, nb, nt, nf, 2 * CH)
                ops.lstm_backward("full", rf, DF, dA, DU, bw[lf.name], lf.hidden, CH)
                g._weight_grads(lf, dA, U, None, F, gview, lambda p: None)
                du = (DU[..., :CH], DU[..., CH:])
                if ctx.needs_input_grad[1]:
                    dfb = torch.empty((nb, nt, nf, CH), dtype=torch.float32, device=dev)
                    train.combine(dfb, plain=du)                                               # through x + fb_skip
                    du = (dfb,)
                if ctx.needs_input_grad[0]:
                    dx = torch.empty((nb, nt, nf, CH), dtype=torch.float32, device=dev)
                    train.combine(dx, masked=dv, plain=du)       # both uses of x (no seed: the "masked" slots are plain sums)
                if dfb is not None:
                    dfb = dfb.view(nb * nt, nf, CH)
        ctx.saved = ctx.bw = None
        grads = tuple(plan.view(grad, n) if ctx.needs_input_grad[5 + i] else None for i, n in enumerate(plan.names))
        return (dx, dfb, None, None, None) + grads


def fnblock_train_forward(blk, x, fb_skip=None):
    """``FNblock.forward`` in train mode: x [nb, nt, nf, C] (+ fb_skip [nb*nt, nf, 256] unless is_first) ->
    (x [nb, nt, nf, Hn], fb_skip [nb*nt, nf, 256], nb_skip [nb*nf, nt, Hn]), differentiable w.r.t. the block's
    parameters and — for blocks 2 / 3 — x and fb_skip."""
    if next(blk.parameters()).dtype != torch.float32:
        raise RuntimeError("FNblock.forward (train mode): fp32 parameters only")
    ops._need_dev(x, fb_skip)
    if not blk.is_first and fb_skip is None:
        raise RuntimeError("FNblock: fb_skip is required unless is_first")
    key = (str(x.device), tuple(tuple(p.shape) for p in blk.parameters()))
    plan = getattr(blk, "_autograd_plan", None)
    if plan is None or blk._autograd_plan_key != key:
        plan = BlockPlan(blk, x.device)
        blk._autograd_plan, blk._autograd_plan_key = plan, key
    base = _next_base(blk)
    layer0 = int(getattr(blk, "dropout_layer", 0))                     # 0 / 2 / 4 inside FN_SSL: layer ids of the masks
    seeds = [train.layer_seed(base, layer0), train.layer_seed(base, layer0 + 1)]
    b0 = getattr(blk, "pair_offset", None)
    b0 = _rank_offset(x.shape[0]) if b0 is None else int(b0)
    if abs(float(blk.dropout) - 0.2) > 1e-12:
        raise RuntimeError("FNblock.forward (train mode): the kernels' keep probability is fixed at 0.8 (dropout=0.2, "
                           "the value every FNblock of the reference is built with, Model.py:9,62-64)")
    fb = fb_skip if not blk.is_first else None
    args = [p for _, p in blk.named_parameters()]
    if fb is None:
        fb = x.new_zeros(())                                           # placeholder (never read, no gradient)
    return FNblockTrainFunction.apply(x, fb, plan, seeds, b0, *args)
