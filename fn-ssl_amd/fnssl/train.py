"""Training step of FN_SSL on one MI355X per rank (BASELINE config 4).

Replaces, for the network part, what Lightning + autograd do in the reference's ``training_step`` /
``cal_loss`` / ``configure_optimizers`` (FN-SSL/Lightning/main.py:149-157, 191-198, 269-271) and DDP's
gradient all-reduce (main.py:286-288):

    forward (train mode)   LSTM kernels that also save their gate activations, dropout + residual adds
                           as one fused element-wise kernel per tensor (fnssl_train_combine)
    loss                   fnssl_mse_loss (MSE of the re-batched prediction)
    backward               fnssl_head_backward, one BPTT kernel per LSTM layer (fnssl_lstm_backward),
                           weight gradients as plain GEMMs  dW = dA^T [x | h_prev]  (rocBLAS via torch.bmm, split-K),
                           gradient accumulation + dropout backward again through fnssl_train_combine
    all-reduce             ONE sum all-reduce of the flat fp32 gradient (2 511 362 floats) over RCCL
    optimizer              fnssl_adam_step on the flat parameter vector (grad / world_size folded in)

PyTorch provides memory, streams, the GEMM library call and torch.distributed; there is no autograd
graph and no CPU fallback.  Dropout masks are a pure function of (seed, step, layer, GLOBAL element index) — the
element index counts pairs over the whole job (rank r's pairs start at r * pairs_per_rank), so an N-rank step
draws exactly the masks the 1-rank step on the concatenated batch draws — and the backward regenerates them
instead of storing them (oracle/train_ref.py restates the hash).  The gradient exchange is asynchronous: each
layer's slice of the flat gradient is all-reduced as soon as its weight gradients are final, i.e. under the
BPTT of the layers below it; the optimizer waits for the last slice.

Tensor naming: logical [pairs, nt, nf, C]; "F" tensors are stored [b, t, f, C] (full-band natural order:
one sequence per (b, t), steps along f), "N" tensors [b, f, t, C] (narrow-band natural order).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import BtfView, check

H_FULL, H_NARR_ONLINE, CH = 128, 256, 256


def _fmix32(h: int) -> int:
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def layer_seed(seed: int, layer: int) -> int:
    """Same as oracle.train_ref.layer_seed."""
    return _fmix32((seed + 0x9E3779B9 * (layer + 1)) & 0xFFFFFFFF)


def _btf(t):
    """logical [b, t, f, C] tensor (any strides, C contiguous) -> BtfView"""
    if t.stride(3) != 1:
        raise RuntimeError("fnssl.train: channel dimension must be contiguous")
    return BtfView(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


@ops.on_device
def combine(out, masked=(), plain=(), seed32=None, b0=0):
    """out = keep_scale(seed32) * sum(masked) + sum(plain); all logical [b, t, f, C]."""
    nb, nt, nf, c = out.shape
    for t in tuple(masked) + tuple(plain):
        if tuple(t.shape) != (nb, nt, nf, c) or not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError("fnssl.train.combine: operand %s vs output %s" % (tuple(t.shape), tuple(out.shape)))
    mv = (BtfView * max(len(masked), 1))(*[_btf(t) for t in masked])
    pv = (BtfView * max(len(plain), 1))(*[_btf(t) for t in plain])
    o = _btf(out)
    check(_lib.load().fnssl_train_combine(o.p, o.sb, o.st, o.sf, nb, nt, nf, c, mv, len(masked), pv, len(plain),
                                          0 if seed32 is None else 1, 0 if seed32 is None else seed32, b0,
                                          ops._stream()), "train_combine")
    return out


def dropout_scale(shape, seed32, device, b0=0):
    """The keep-scale tensor itself, logical [b, t, f, C] contiguous (tests / debugging)."""
    nb, nt, nf, c = shape
    out = torch.empty(shape, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        check(_lib.load().fnssl_dropout_scale(out.data_ptr(), out.numel(), seed32, b0 * nt * nf * c, ops._stream()),
              "dropout_scale")
    return out


def sync_gradients(flat_grad, group=None):
    """DDP's exchange step (main.py:286-288): ONE sum all-reduce of the flat gradient; returns the factor the
    optimizer has to apply (1 / world_size — Lightning DDP averages).  Works on any backend (RCCL on the
    GPUs; the CPU tests run it over gloo)."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def shard_utterances(n_utts: int, rank: int, world: int):
    """Contiguous utterance range of this rank (a global batch of 256 -> 32 per GPU at world 8)."""
    per, rem = divmod(n_utts, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


class _Layer:
    """Static description of one LSTM of the network."""

    def __init__(self, name, mode, hidden, ndir, c0, c2, c0g):
        self.name, self.mode, self.hidden, self.ndir, self.c0, self.c2, self.c0g = name, mode, hidden, ndir, c0, c2, c0g
        self.sfx = [""] + (["_reverse"] if ndir == 2 else [])


def build_layers(online: bool):
    """The six LSTMs of FN_SSL (Model.py:25-29) in forward order: full 1, narrow 1, full 2, ..."""
    nh = H_NARR_ONLINE if online else H_FULL
    nd = 1 if online else 2
    layers = []
    for k in (1, 2, 3):
        first = k == 1
        layers.append(_Layer("block_%d.fullLstm" % k, "full", H_FULL, 2, 4 if first else CH, 0, 0 if first else CH))
        layers.append(_Layer("block_%d.narrLstm" % k, "narrow", nh, nd, CH, 4 if first else 0, CH))
    return layers


def flat_layout(named_shapes):
    """name -> (offset, shape) of a flat fp32 vector holding the parameters in the given order; element 0 is a constant
    0 so that index maps can point "nowhere".  Returns (offset dict, total length incl. element 0)."""
    offset, off = {}, 1
    for k, shape in named_shapes:
        offset[k] = (off, tuple(shape))
        off += int(np.prod(shape))
    return offset, off


def build_index_maps(layers, offset, dev):
    """Gather maps flat-vector -> packed weight streams, built once by running the host packers on
    index-valued weights (indices < 2^24 are exact in fp32)."""
    maps = {}
    for L in layers:
        for s in L.sfx:
            oi, shi = offset["%s.weight_ih_l0%s" % (L.name, s)]
            oh, shh = offset["%s.weight_hh_l0%s" % (L.name, s)]
            obi, shb = offset["%s.bias_ih_l0%s" % (L.name, s)]
            obh, _ = offset["%s.bias_hh_l0%s" % (L.name, s)]
            wi = (oi + np.arange(np.prod(shi), dtype=np.float64)).astype(np.float32).reshape(shi)
            wh = (oh + np.arange(np.prod(shh), dtype=np.float64)).astype(np.float32).reshape(shh)
            bi = (obi + np.arange(shb[0], dtype=np.float64)).astype(np.float32)
            bh = (obh + np.arange(shb[0], dtype=np.float64)).astype(np.float32)
            z = np.zeros(shb[0], dtype=np.float32)
            a = ops.pack_lstm_host(wi, wh, bi, z, L.c0, L.c2)
            b = ops.pack_lstm_host(np.zeros_like(wi), np.zeros_like(wh), z, bh, L.c0, L.c2)
            bw = ops.pack_lstm_bwd_host(wi, wh, L.c0g)
            to = lambda v: torch.from_numpy(v.astype(np.int64)).to(dev)  # noqa: E731
            maps[(L.name, s)] = (to(a), to(b), to(bw))
    return maps


def pack_streams(layers, maps, theta):
    """Device-side re-pack of every weight stream from the current flat parameters (two gathers per direction)."""
    fw, bw = {}, {}
    for L in layers:
        fw[L.name], bw[L.name] = [], []
        for s in L.sfx:
            ia, ib, ibw = maps[(L.name, s)]
            fw[L.name].append(theta[ia] + theta[ib])
            bw[L.name].append(theta[ibw])
    return fw, bw


class TrainGraph:
    """Train-mode forward and backward of FN_SSL (Model.py:31-50, 72-90) as two halves over the C-ABI kernels.  Shared by
    ``TrainEngine`` (which runs loss + both halves per chunk, buffers reused across steps) and ``fnssl.autograd`` (forward
    in ``Function.forward``, backward in ``Function.backward``, fresh buffers per call).

    ``alloc(key, shape)`` returns an fp32 device tensor; ``pview(name)`` a parameter, ``gview(name)`` the tensor its
    gradient is ACCUMULATED into; ``layer_done(prefix)`` is called when every gradient under ``prefix`` is final.
    ``seeds``: six 32-bit dropout seeds (``layer_seed``) or None = dropout off (the eval-mode graph)."""

    def __init__(self, layers, alloc, fallbacks=None, bwd_tuning=None):
        self.layers, self.alloc = layers, alloc
        # optional int32 device counter handed to every LSTM call: layers whose cluster-resident kernel gave up on a
        # hand-off and were recomputed by the guarded fallback kernels of the same call (correct, but slow)
        self.fallbacks = fallbacks
        # optional per-call tuning of the BACKWARD's LSTM kernels (``_lib.make_tuning(reserved_cus=16)``: the compute units
        # RCCL's all-reduce kernels hold while the BPTT of the layers below runs)
        self.bwd_tuning = bwd_tuning
        self.Lf = [layers[0], layers[2], layers[4]]
        self.Ln = [layers[1], layers[3], layers[5]]

    def _natural(self, key, L, nbp, nt, nf, c):
        """Logical [b, t, f, c] tensor stored in layer L's natural layout."""
        if L.mode == "full":
            return self.alloc(key, (nbp, nt, nf, c))
        return self.alloc(key, (nbp, nf, nt, c)).permute(0, 2, 1, 3)

    @staticmethod
    def _rows(t_logical, L):
        """[rows = seq * step, C] matrix of a tensor stored in L's natural layout."""
        st = t_logical if L.mode == "full" else t_logical.permute(0, 2, 1, 3)
        return st.reshape(-1, st.shape[-1])

    def _weight_grads(self, L, da, x0, x2, hout, gview, layer_done):
        """dW_ih = dA^T [x0 | x2],  dW_hh = dA^T h_prev,  db = sum dA, accumulated into the gradient tensors: ONE
        hand-written split-K fp32-MFMA launch per layer (``fnssl_lstm_weight_grads``, csrc/wgrad.hip; both directions, the
        three operand segments read in place, h_prev as an index shift) — no vendor GEMM on the path."""
        nsteps = da.shape[2] if L.mode == "full" else da.shape[1]
        g = lambda n: [gview("%s.%s%s" % (L.name, n, s)) for s in L.sfx]   # noqa: E731
        ops.lstm_weight_grads(self._rows(da, L), self._rows(x0, L) if L.c0 else None, self._rows(x2, L) if L.c2 else None,
                              self._rows(hout, L), L.hidden, L.ndir, nsteps, g("weight_ih_l0"), g("weight_hh_l0"),
                              g("bias_ih_l0"), g("bias_hh_l0"))
        layer_done(L.name + ".")

    def forward(self, x, fw, seeds, b0, pview):
        """x [nbp, 4, nf, nt] features -> (pred [nbp, nt//12, 2nf], saved activations for ``backward``)."""
        nbp, _, nf, nt = x.shape
        Lf, Ln = self.Lf, self.Ln
        sd = (lambda i: None) if seeds is None else (lambda i: seeds[i])
        XF = ops.nchw_to_seq(x)                                    # [b, t, f, 4]
        XN = XF.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)   # same numbers, stored [b, f, t, 4]
        res, U, F, V, N = {}, {}, {}, {}, {}
        Xk = None
        for k in (1, 2, 3):
            lf, ln = Lf[k - 1], Ln[k - 1]
            F[k] = self._natural("F%d" % k, lf, nbp, nt, nf, 2 * H_FULL)
            res[lf.name] = self.alloc("R" + lf.name, (ops.lstm_reserve_floats(nbp * nt, lf.hidden, 2, nf),))
            if k == 1:
                ops.lstm_layer("full", XF, None, None, fw[lf.name], lf.hidden, F[k], reserve=res[lf.name], fallback_count=self.fallbacks)
            else:
                U[k] = self._natural("U%d" % k, lf, nbp, nt, nf, CH)
                combine(U[k], plain=(Xk, F[k - 1]))                                   # x + fb_skip  (:36-37)
                ops.lstm_layer("full", U[k], None, None, fw[lf.name], lf.hidden, F[k], reserve=res[lf.name],
                               fallback_count=self.fallbacks)
            V[k] = self._natural("V%d" % k, ln, nbp, nt, nf, CH)
            if k == 1:
                combine(V[k], masked=(F[k],), seed32=sd(0), b0=b0)                   # dropout_full (:40)
            else:
                combine(V[k], masked=(F[k],), plain=(Xk,), seed32=sd(2 * k - 2), b0=b0)   # + nb_skip (:44-45)
            N[k] = self._natural("N%d" % k, ln, nbp, nt, nf, ln.ndir * ln.hidden)
            res[ln.name] = self.alloc("R" + ln.name, (ops.lstm_reserve_floats(nbp * nf, ln.hidden, ln.ndir, nt),))
            ops.lstm_layer("narrow", V[k], None, XN if k == 1 else None, fw[ln.name], ln.hidden, N[k],
                           reserve=res[ln.name], fallback_count=self.fallbacks)
            Xk = self._natural("X", ln, nbp, nt, nf, CH)
            combine(Xk, masked=(N[k],), seed32=sd(2 * k - 1), b0=b0)                # dropout_narr (:48)
        X4 = Xk.permute(0, 2, 1, 3)                                                  # storage [b, f, t, 256]
        pred = ops.head(X4, pview("emb2ipd.weight"), pview("emb2ipd.bias"))          # [nbp, nt2, 2nf]
        saved = {"XF": XF, "XN": XN, "res": res, "U": U, "F": F, "V": V, "N": N, "X4": X4, "pred": pred,
                 "shape": (nbp, nf, nt)}
        return pred, saved

    def backward(self, saved, dpred, bw, seeds, b0, pview, gview, layer_done=lambda prefix: None):
        """Accumulate every parameter gradient of the chunk given dL/dpred (contiguous [nbp, nt//12, 2nf])."""
        nbp, nf, nt = saved["shape"]
        Lf, Ln = self.Lf, self.Ln
        sd = (lambda i: None) if seeds is None else (lambda i: seeds[i])
        XF, XN, res, U, F, V, N, X4, pred = (saved[k] for k in ("XF", "XN", "res", "U", "F", "V", "N", "X4", "pred"))
        lib = _lib.load()
        bk = {"fallback_count": self.fallbacks, "tuning": self.bwd_tuning}
        wname, bname = "emb2ipd.weight", "emb2ipd.bias"
        ws = self.alloc("ws_small", (max(lib.fnssl_head_backward_workspace_bytes() // 4, 256),))
        G = self.alloc("G", (nbp, nf, nt, CH))                                       # dL/dX4, N storage
        check(lib.fnssl_head_backward(X4.data_ptr(), pview(wname).data_ptr(), pred.data_ptr(), dpred.data_ptr(),
                                      nbp, nf, nt, G.data_ptr(), gview(wname).data_ptr(),
                                      gview(bname).data_ptr(), 1, ws.data_ptr(), ws.numel() * 4, ops._stream()),
              "head_backward")
        layer_done("emb2ipd.")
        gx = (G.permute(0, 2, 1, 3),)                     # operands whose sum is dL/dX_{k+1}
        dfb = ()                                           # operands whose sum is dL/dF_k through fb_skip
        for k in (3, 2, 1):
            lf, ln = Lf[k - 1], Ln[k - 1]
            DN = self._natural("DN", ln, nbp, nt, nf, ln.ndir * ln.hidden)
            combine(DN, masked=gx, seed32=sd(2 * k - 1), b0=b0)                      # dropout_narr backward
            dA = self._natural("dA", ln, nbp, nt, nf, ln.ndir * 4 * ln.hidden)
            DV = self._natural("DV%d" % (k & 1), ln, nbp, nt, nf, ln.ndir * CH)
            ops.lstm_backward("narrow", res[ln.name], DN, dA, DV, bw[ln.name], ln.hidden, CH, **bk)
            self._weight_grads(ln, dA, V[k], XN if k == 1 else None, N[k], gview, layer_done)
            dv = tuple(DV[..., d * CH:(d + 1) * CH] for d in range(ln.ndir))        # one slab per direction
            DF = self._natural("DF", lf, nbp, nt, nf, 2 * H_FULL)
            combine(DF, masked=dv, plain=dfb, seed32=sd(2 * k - 2), b0=b0)          # dropout_full backward + fb_skip
            dA = self._natural("dA", lf, nbp, nt, nf, 2 * 4 * H_FULL)
            if k > 1:
                DU = self._natural("DU%d" % (k & 1), lf, nbp, nt, nf, 2 * CH)
                ops.lstm_backward("full", res[lf.name], DF, dA, DU, bw[lf.name], lf.hidden, CH, **bk)
                self._weight_grads(lf, dA, U[k], None, F[k], gview, layer_done)
                du = (DU[..., :CH], DU[..., CH:])
                if len(dv) + 2 > 3:          # offline narrow-band: 2 + 2 operands -> fold the full-band pair first
                    S = self._natural("S", lf, nbp, nt, nf, CH)
                    combine(S, plain=du)
                    du = (S,)
                gx = dv + du                 # dL/dX_k = dV_k + dU_k  (both uses of x: nb_skip and the full-band input)
                dfb = du                     # dL/dF_{k-1} through fb_skip
            else:
                ops.lstm_backward("full", res[lf.name], DF, dA, None, bw[lf.name], lf.hidden, 0, **bk)
                self._weight_grads(lf, dA, XF, None, F[k], gview, layer_done)


class TrainEngine:
    """Owns the flat parameter / gradient / Adam-moment vectors of a ``Model.FN_SSL`` and runs training steps.

    ``model`` must live on a ROCm device; its parameters are re-pointed into the flat vector (so
    ``state_dict()`` / checkpoints keep working and always show the current weights).
    ``chunk_pairs`` bounds the activation memory: the rank's batch is processed in chunks of that many
    microphone pairs with gradient accumulation (the result does not depend on it).
    """

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, chunk_pairs=None, process_group=None):
        if getattr(model, "is_doa", False):
            raise RuntimeError("fnssl.train: the DOA-classification variant is not part of the training path")
        self.model = model
        self.lr, self.betas, self.eps, self.seed = lr, betas, eps, seed
        self.chunk_pairs = chunk_pairs
        self.pg = process_group          # None: the default group (if initialised); False: never distributed
        self.force_seed = None           # tests / bench parity: use exactly this base seed for the next steps
        self._pending, self._wait_events, self._reduce_now = [], None, False
        self.last_comm_launches = 0      # all-reduces issued by the last step (one per layer with more than one rank)
        self.step_count = 0
        self.online = bool(model.is_online)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("fnssl.train: the model must be on a ROCm device (no CPU path)")
        self.dev = dev
        named = list(model.named_parameters())
        self.names = [k for k, _ in named]
        self.offset, total = flat_layout([(k, p.shape) for k, p in named])
        self.nparam = total - 1
        # element 0 is a constant 0 so that index maps can point "nowhere"
        self.theta = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.theta)
        self.exp_avg = torch.zeros_like(self.theta)
        self.exp_avg_sq = torch.zeros_like(self.theta)
        for k, p in named:
            off, shape = self.offset[k]
            n = p.numel()
            self.theta[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.theta[off:off + n].view(p.shape)
        self.layers = build_layers(self.online)
        self.maps = build_index_maps(self.layers, self.offset, dev)
        self.loss_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = {}
        # LSTM launches (forward and backward) whose cluster-resident kernel gave up and were recomputed by the guarded
        # fallback kernels: a device counter, never read on the step path (``cluster_fallbacks()`` synchronises)
        self.fallbacks = ops.fallback_counter(dev)
        # compute units RCCL's all-reduce kernels hold while the BPTT of the lower layers runs (the per-layer asynchronous
        # all-reduce): with more than one rank the backward's cluster kernels size their co-resident grids for the rest
        self.reserved_cus = 16
        self.reserve_always = False      # tests: reserve with one rank too
        self.reduce_single_rank = False  # tests: issue the per-layer all-reduces on a ONE-rank group too (a sum over one
                                         # rank is the identity: it exercises the RCCL plumbing a single GPU can show)

    # ------------------------------------------------------------------ distributed plumbing
    def _world_rank(self):
        dist = torch.distributed
        if self.pg is False or not (dist.is_available() and dist.is_initialized()):
            return 1, 0
        return dist.get_world_size(self.pg), dist.get_rank(self.pg)

    def global_pair_offset(self, nbp: int) -> int:
        """Index of this rank's first pair in the global batch = the pairs of all lower ranks.  Equal shards give
        rank * nbp without communication; shards are compared with ONE small all-gather (shard_utterances() hands
        the first ranks one utterance more when the batch does not divide), so uneven shards never draw overlapping
        dropout masks.  This is a collective plus a host read-back: callers whose shards are equal by construction
        (a DistributedSampler, ``equal_shard_pair_offset``) pass ``pair_offset`` to ``step`` and never get here."""
        world, rank = self._world_rank()
        if world <= 1:
            return 0
        dist = torch.distributed
        mine = torch.tensor([int(nbp)], dtype=torch.int64,
                            device=self.dev if dist.get_backend(self.pg or None) == "nccl" else "cpu")
        counts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(counts, mine, group=self.pg or None)
        return int(sum(int(c.item()) for c in counts[:rank]))

    def equal_shard_pair_offset(self, nbp: int) -> int:
        """rank * nbp: the first global pair of this rank when every rank holds ``nbp`` pairs (what DDP's
        DistributedSampler guarantees) — no communication, no host synchronisation."""
        _, rank = self._world_rank()
        return rank * int(nbp)

    def _bucket(self, prefix):
        """[lo, hi) of the flat vector covered by the parameters whose name starts with ``prefix`` (contiguous:
        the flat order is named_parameters order)."""
        offs = [(self.offset[k][0], int(np.prod(self.offset[k][1]))) for k in self.names if k.startswith(prefix)]
        return min(o for o, _ in offs), max(o + n for o, n in offs)

    def _reduce_async(self, prefix):
        """Start the sum all-reduce of one layer's gradient slice (RCCL runs it on its own stream, ordered after
        everything enqueued so far on the compute stream)."""
        world, _ = self._world_rank()
        if not self._reduce_now or (world <= 1 and not (self.reduce_single_rank and self.pg is not False and
                                                        torch.distributed.is_available() and torch.distributed.is_initialized())):
            return
        lo, hi = self._bucket(prefix)
        self.last_comm_launches += 1
        self._pending.append(torch.distributed.all_reduce(self.grad[lo:hi], op=torch.distributed.ReduceOp.SUM,
                                                          group=self.pg or None, async_op=True))

    def _wait_reductions(self):
        if not self._pending:
            self._wait_events = None
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for w in self._pending:
            w.wait()
        e1.record()
        self._pending, self._wait_events = [], (e0, e1)

    @property
    def last_comm_wait_ms(self):
        """GPU time the compute stream spent waiting for the gradient all-reduces of the last step."""
        if self._wait_events is None:
            return 0.0
        self._wait_events[1].synchronize()
        return float(self._wait_events[0].elapsed_time(self._wait_events[1]))

    # ------------------------------------------------------------------ parameter plumbing
    def gview(self, name):
        off, shape = self.offset[name]
        return self.grad[off:off + int(np.prod(shape))].view(shape)

    def pview(self, name):
        off, shape = self.offset[name]
        return self.theta[off:off + int(np.prod(shape))].view(shape)

    def _pack_all(self):
        """Device-side re-pack of every weight stream from the current flat parameters (two gathers)."""
        return pack_streams(self.layers, self.maps, self.theta)

    def _buf(self, key, shape):
        n = int(np.prod(shape))
        b = self._scratch.get(key)
        if b is None or b.numel() < n:
            b = torch.empty(n, dtype=torch.float32, device=self.dev)
            self._scratch[key] = b
        return b[:n].view(shape)

    # ------------------------------------------------------------------ one chunk of pairs
    def _layer_done(self, prefix):
        self._reduce_async(prefix)

    def _backward_tuning(self):
        """Per-call tuning of the backward's LSTM kernels: RESERVED_CUS when gradients are all-reduced under the BPTT."""
        world, _ = self._world_rank()
        if not self.reserved_cus or (world <= 1 and not self.reserve_always):
            return None
        return _lib.make_tuning(reserved_cus=self.reserved_cus)      # on top of the current default knobs

    def cluster_fallbacks(self) -> int:
        """LSTM launches so far that fell back from a cluster-resident kernel to the per-wave / split kernels (synchronises)."""
        return int(self.fallbacks.item())

    def _chunk(self, x, gt, b0, n_total, fw, bw, seeds):
        """Forward + loss + backward of pairs [b0, b0 + nbp) (x [nbp, 4, nf, nt]); accumulates grads and the loss."""
        nbp, _, nf, nt = x.shape
        npair = gt.shape[3]
        graph = TrainGraph(self.layers, self._buf, self.fallbacks, self._backward_tuning())
        pred, saved = graph.forward(x, fw, seeds, b0, self.pview)
        nt2 = pred.shape[1]
        lib = _lib.load()
        dpred = self._buf("dpred", tuple(pred.shape))
        ws = self._buf("ws_small", (max(lib.fnssl_head_backward_workspace_bytes() // 4, 256),))
        check(lib.fnssl_mse_loss(pred.data_ptr(), gt.data_ptr(), nbp // npair, npair, nt2, 2 * nf, n_total,
                                 dpred.data_ptr(), self.loss_dev.data_ptr(), 1, ws.data_ptr(), ws.numel() * 4,
                                 ops._stream()), "mse_loss")
        graph.backward(saved, dpred, bw, seeds, b0, self.pview, self.gview, self._layer_done)
        return pred

    # ------------------------------------------------------------------ public API
    @ops.on_device
    def step(self, x, gt_ipd, sync_loss=True, pair_offset=None):
        """One optimisation step.  x [nb*np, 4, nf, nt] features (data_preprocess output), gt_ipd
        [nb, nt//12, 2*nf, np] targets, both on the device.  Returns the loss of THIS rank's shard (float, or the
        device scalar when ``sync_loss`` is False).  ``pair_offset``: index of this rank's first pair in the global
        batch (default rank * pairs_per_rank, i.e. equal contiguous shards) — it only keys the dropout masks."""
        ops._need_dev(x, gt_ipd)
        nbp, cin, nf, nt = x.shape
        nb, nt2, nf2, npair = gt_ipd.shape
        if cin != 4 or nb * npair != nbp or nt2 != nt // 12 or nf2 != 2 * nf or nt2 == 0:
            raise RuntimeError("fnssl.train.step: x %s does not match gt_ipd %s" % (tuple(x.shape), tuple(gt_ipd.shape)))
        self.step_count += 1
        world, rank = self._world_rank()
        base = (self.seed * 1000003 + self.step_count * 8191) & 0xFFFFFFFF if self.force_seed is None else self.force_seed
        if pair_offset is None:
            pair_offset = self.global_pair_offset(nbp)
        pair0 = int(pair_offset)                                              # this rank's first GLOBAL pair index
        seeds = [layer_seed(base, l) for l in range(6)]
        self.last_seed = base
        fw, bw = self._pack_all()
        self.grad.zero_()
        self.loss_dev.zero_()
        self.last_comm_launches = 0
        n_total = nbp * nt2 * nf2
        cp = self.chunk_pairs or nbp
        cp = max(npair, (cp // npair) * npair)            # whole utterances per chunk
        gt_ipd = gt_ipd.contiguous()
        try:
            for b0 in range(0, nbp, cp):
                b1 = min(nbp, b0 + cp)
                self._reduce_now = b1 == nbp          # gradients are final only in the last chunk's backward
                self._chunk(x[b0:b1].contiguous(), gt_ipd[b0 // npair:b1 // npair], pair0 + b0, n_total, fw, bw, seeds)
            self._reduce_now = False
            self._wait_reductions()
        except BaseException:
            # a failed chunk (OOM, HIP error) must not leave stale collective handles or the "reduce now" flag behind
            self._reduce_now = False
            for w in self._pending:
                try:
                    w.wait()
                except Exception:
                    pass
            self._pending, self._wait_events = [], None
            raise
        gscale = 1.0 / world
        check(_lib.load().fnssl_adam_step(self.theta.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                          self.exp_avg_sq.data_ptr(), self.theta.numel(), self.lr, self.betas[0],
                                          self.betas[1], self.eps, self.step_count, gscale, ops._stream()),
              "adam_step")
        self.theta[0] = 0.0
        # the parameters changed under the inference modules' feet: drop their packed-weight caches
        for m in self.model.modules():
            for attr in ("_packed_key", "_net_key"):
                if hasattr(m, attr):
                    setattr(m, attr, None)
        return float(self.loss_dev.item()) if sync_loss else self.loss_dev

    def gradients(self):
        """name -> gradient tensor (views of the flat vector; as left by the last step, before the 1/world scale)."""
        return {k: self.gview(k) for k in self.names}


class CTrainStep:
    """The training backward / step as ONE C call each (``fnssl_train_backward`` / ``fnssl_train_step``): forward in
    train mode, loss, BPTT, weight gradients (rocBLAS GEMMs called from the library), Adam — no Python in between.
    Shares the flat parameter / gradient / moment vectors of a ``TrainEngine`` (whose layout it verifies against the
    library's own table), so the two paths can be compared and mixed; the Python-orchestrated engine remains the one
    that overlaps the per-layer gradient all-reduce with the backward."""

    def __init__(self, engine: "TrainEngine"):
        self.eng = engine
        lib = _lib.load()
        h = C.c_void_p()
        check(lib.fnssl_train_create(1 if engine.online else 0, C.byref(h)), "train_create")
        self.h = h
        if lib.fnssl_train_param_floats(h) != engine.theta.numel():
            raise RuntimeError("fnssl.train.CTrainStep: flat vector length %d != library's %d"
                               % (engine.theta.numel(), lib.fnssl_train_param_floats(h)))
        for li, L in enumerate(engine.layers):
            for di, s in enumerate(L.sfx):
                for wi, n in enumerate(("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")):
                    if lib.fnssl_train_param_offset(h, li, di, wi) != engine.offset["%s.%s%s" % (L.name, n, s)][0]:
                        raise RuntimeError("fnssl.train.CTrainStep: parameter layout mismatch at %s.%s%s" % (L.name, n, s))
        if (lib.fnssl_train_param_offset(h, 6, 0, 0) != engine.offset["emb2ipd.weight"][0] or
                lib.fnssl_train_param_offset(h, 6, 0, 1) != engine.offset["emb2ipd.bias"][0]):
            raise RuntimeError("fnssl.train.CTrainStep: parameter layout mismatch at emb2ipd")
        self.maps = torch.empty(lib.fnssl_train_map_bytes(h), dtype=torch.uint8, device=engine.dev)
        with torch.cuda.device(engine.dev):
            check(lib.fnssl_train_upload_maps(h, self.maps.data_ptr(), ops._stream()), "train_upload_maps")
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.load().fnssl_train_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _workspace(self, nbp, nf, nt):
        n = _lib.load().fnssl_train_workspace_bytes(self.h, nbp, nf, nt)
        if self._ws is None or self._ws.numel() < n:
            self._ws = None
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.eng.dev)
        return self._ws

    @ops.on_device
    def backward(self, x, gt_ipd, seed_base, pair0=0, n_total=None):
        """Accumulate loss and gradients of one chunk into the engine's ``loss_dev`` / ``grad`` (zero them first)."""
        e = self.eng
        ops._need_dev(x, gt_ipd)
        x, gt_ipd = x.contiguous(), gt_ipd.contiguous()
        nbp, _, nf, nt = x.shape
        nb, nt2, nf2, npair = gt_ipd.shape
        if n_total is None:
            n_total = nbp * nt2 * nf2
        ws = self._workspace(nbp, nf, nt)
        check(_lib.load().fnssl_train_backward(self.h, e.theta.data_ptr(), e.grad.data_ptr(), x.data_ptr(), gt_ipd.data_ptr(),
                                               nb, npair, nf, nt, seed_base & 0xFFFFFFFF, pair0, n_total,
                                               e.loss_dev.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
              "train_backward")

    def step(self, x, gt_ipd, seed_base):
        """One single-process optimisation step (zero, backward, Adam) in one C call; returns the loss."""
        self.step_nosync(x, gt_ipd, seed_base)
        return float(self.eng.loss_dev.item())

    @ops.on_device
    def step_nosync(self, x, gt_ipd, seed_base):
        """``step`` without reading the loss back (it stays in ``engine.loss_dev``)."""
        e = self.eng
        ops._need_dev(x, gt_ipd)
        x, gt_ipd = x.contiguous(), gt_ipd.contiguous()
        nbp, _, nf, nt = x.shape
        nb, nt2, nf2, npair = gt_ipd.shape
        ws = self._workspace(nbp, nf, nt)
        e.step_count += 1
        check(_lib.load().fnssl_train_step(self.h, e.theta.data_ptr(), e.grad.data_ptr(), e.exp_avg.data_ptr(),
                                           e.exp_avg_sq.data_ptr(), x.data_ptr(), gt_ipd.data_ptr(), nb, npair, nf, nt,
                                           seed_base & 0xFFFFFFFF, e.lr, e.betas[0], e.betas[1], e.eps, e.step_count,
                                           e.loss_dev.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()), "train_step")
        for m in e.model.modules():
            for attr in ("_packed_key", "_net_key"):
                if hasattr(m, attr):
                    setattr(m, attr, None)
