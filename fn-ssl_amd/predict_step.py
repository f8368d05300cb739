"""The Lightning predict entry of the reference (FN-SSL/Lightning/main.py:81-134,184-225),
restated over the HIP path: ``MyModel.predict_step(batch[nb, nch, ns], batch_idx)``
returns the raw DP-IPD predictions ``[nb*np, nt//12, 512]``.

``pytorch_lightning`` is optional: with it installed ``MyModel`` is a
``LightningModule`` (so ``Trainer.predict`` / ``LightningCLI`` drive it exactly
like the reference's ``main.py predict``); without it the same class derives
from ``nn.Module`` and ``predict_step`` is called directly (``Predict.py``).
``training_step`` (main.py:149-157) has two routes, chosen by the constructor's ``fused_engine``:

* ``fused_engine=False`` — the reference's own code path: ``pred = self(in_batch)`` in train mode carries an autograd
  graph (fnssl/autograd.py: ``torch.autograd.Function`` over the reserve-saving forward and the BPTT / weight-gradient
  kernels), ``cal_loss`` is differentiable, the returned ``{"loss": loss}`` is what Lightning's AUTOMATIC optimisation
  calls ``backward()`` on, ``configure_optimizers`` returns the reference's ``torch.optim.Adam`` + ``ExponentialLR``
  (main.py:269-279), and under ``strategy="ddp"`` (main.py:286-288) DDP's reducer all-reduces the ``.grad``s over RCCL.
* ``fused_engine=True`` (default; the fast path) — the HIP training engine (fnssl.train: forward with dropout, MSE loss,
  BPTT, per-layer asynchronous RCCL gradient all-reduce, fused Adam) carries its own backward and optimizer and returns
  the loss as a detached scalar; ``configure_optimizers`` hands Lightning an ``EngineOptimizer`` — a
  ``torch.optim.Optimizer`` whose ``step()`` does no arithmetic (the engine has already applied Adam) but which
  Lightning's manual-optimisation loop counts, so ``trainer.global_step`` advances and ``ModelCheckpoint`` /
  ``max_steps`` / logger step indices work, and whose ``state_dict`` carries the engine's Adam moments into checkpoints.
  The numpy DP-IPD target generator and the DOA metrics stay outside this path
(SURVEY.md §8): ``gt_batch['ipd']`` must already hold the target IPDs.
"""
import torch

import Model as at_model
from fnssl import ops

try:  # optional, absent in the build image
    from pytorch_lightning import LightningModule as _Base
except Exception:  # pragma: no cover
    _Base = torch.nn.Module


class EngineOptimizer(torch.optim.Optimizer):
    """What ``configure_optimizers`` returns.  The HIP engine owns the parameters' flat copy, the gradients and the Adam
    moments and applies the update inside ``TrainEngine.step``; this object exists for the trainer's bookkeeping:
    ``step()`` is counted by Lightning (``optim_step_progress`` -> ``trainer.global_step``), ``state_dict()`` /
    ``load_state_dict()`` move the engine's optimizer state (moments, step count, learning rate) through checkpoints."""

    def __init__(self, module):
        self._module = module
        self._anchor = torch.nn.Parameter(torch.zeros(1), requires_grad=False)     # Optimizer needs >= 1 parameter
        super().__init__([self._anchor], {"lr": 0.001})
        self.steps_taken = 0

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.steps_taken += 1
        return loss

    def zero_grad(self, set_to_none: bool = True):
        return None                                   # the engine zeroes its flat gradient at the start of every step

    def state_dict(self):
        eng = getattr(self._module, "_train_engine", None)
        sd = {"steps_taken": self.steps_taken, "engine": None}
        if eng is not None:
            sd["engine"] = {"exp_avg": eng.exp_avg.detach().cpu(), "exp_avg_sq": eng.exp_avg_sq.detach().cpu(),
                            "step_count": int(eng.step_count), "lr": float(eng.lr)}
        return sd

    def load_state_dict(self, sd):
        self.steps_taken = int(sd.get("steps_taken", 0))
        es = sd.get("engine")
        if es is not None:
            eng = self._module._engine()
            eng.exp_avg.copy_(es["exp_avg"].to(eng.exp_avg.device))
            eng.exp_avg_sq.copy_(es["exp_avg_sq"].to(eng.exp_avg_sq.device))
            eng.step_count, eng.lr = int(es["step_count"]), float(es["lr"])


class _MSELoss(torch.autograd.Function):
    """cal_loss (main.py:191-198) as one HIP kernel with its gradient: loss = mean((rebatch(pred) - gt)^2)."""

    @staticmethod
    def forward(ctx, pred, gt):
        pred, gt = pred.contiguous(), gt.contiguous()
        nb, nt2, nf2, npair = gt.shape
        with torch.cuda.device(pred.device):
            dpred = torch.empty_like(pred)
            loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
            ws = torch.empty(256, dtype=torch.float32, device=pred.device)
            ops.check(ops._lib.load().fnssl_mse_loss(pred.data_ptr(), gt.data_ptr(), nb, npair, nt2, nf2, pred.numel(),
                                                     dpred.data_ptr(), loss.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4,
                                                     ops._stream()), "mse_loss")
        ctx.dpred = dpred
        return loss.reshape(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return ctx.dpred * g, None


class MyModel(_Base):
    def __init__(self, tar_useVAD: bool = True, ch_mode: str = 'MM', res_the: int = 37, res_phi: int = 73,
                 fs: int = 16000, win_len: int = 512, nfft: int = 512, win_shift_ratio: float = 0.5,
                 method_mode: str = 'IDL', source_num_mode: str = 'KNum', max_num_sources: int = 1,
                 return_metric: bool = True, exp_name: str = 'exp', compile: bool = False,
                 device: str = "cuda", fused_engine: bool = True, mic_location=None):
        super().__init__()
        if (win_len, nfft, win_shift_ratio) != (512, 512, 0.5):
            raise ValueError("the MI355X path is built for win_len = nfft = 512, hop 256 (main.py:38-44)")
        self.arch = at_model.FN_SSL()
        self.fused_engine = bool(fused_engine)
        # fused engine: the optimisation runs inside the HIP engine (training_step = one complete step and returns a
        # detached loss): manual optimisation, or Lightning 2.x would call backward() on that loss.  Otherwise the
        # reference's automatic optimisation: Lightning calls loss.backward() (-> fnssl.autograd) and optimizer.step().
        if hasattr(self, "automatic_optimization") or _Base is not torch.nn.Module:
            self.automatic_optimization = not self.fused_engine
        self.method_mode, self.source_num_mode, self.max_num_sources = method_mode, source_num_mode, max_num_sources
        self.ch_mode = ch_mode
        self.tar_useVAD = tar_useVAD
        self.fre_max = fs / 2
        # the array geometry the DP-IPD targets are generated for: the reference hard-codes its two-microphone array
        # (main.py:121-123); other arrays pass theirs
        import numpy as _np
        self.mic_location = _np.array(((-0.04, 0.0, 0.0), (0.04, 0.0, 0.0))) if mic_location is None else _np.asarray(mic_location)
        self.speed = 340
        self.nfft = nfft
        self.dev = device
        self.fre_range_used = range(1, int(self.nfft / 2) + 1, 1)
        self.eval()

    def forward(self, x):
        return self.arch(x)

    # ---- training (main.py:149-157, 191-198, 269-279) ------------------------------------------
    def _engine(self):
        if getattr(self, "_train_engine", None) is None:
            from fnssl import train
            self._train_engine = train.TrainEngine(self.arch, lr=0.001)          # Adam, lr 1e-3 (main.py:270)
        return self._train_engine

    def training_step(self, batch, batch_idx: int = 0):
        """batch = (mic_sig_batch [nb, ns, nch], gt_batch with 'ipd' [nb, nt//12, 512, np]).
        One complete optimisation step on this rank's shard; returns {"loss": detached scalar}."""
        mic_sig_batch, gt_batch = batch[0], batch[1]
        if 'ipd' not in gt_batch:          # the reference's batches carry DOAs and VADs: main.py:152 builds the targets here
            x, gt_batch = self.data_preprocess(mic_sig_batch, gt_batch)
        else:
            x = ops.preprocess(mic_sig_batch.to(self.dev), self.ch_mode, 1e-6, layout=1)
        if not self.fused_engine:
            # main.py:153-157 as written: forward with a graph (the module must be in train() mode, which Lightning's fit
            # loop / Learner.train_epoch set), differentiable loss; Lightning or the caller does backward + optimizer step
            pred_batch = self(x)
            loss = self.cal_loss(pred_batch=pred_batch, gt_batch=gt_batch)
            if hasattr(self, "log") and getattr(self, "_trainer", None) is not None:
                self.log("train/loss", loss, prog_bar=True)
            return {"loss": loss}
        eng = self._engine()
        # a DistributedSampler hands every rank the same number of utterances: the first global pair of this rank is
        # rank * pairs, no collective and no host synchronisation per step
        loss = eng.step(x, gt_batch['ipd'].to(self.dev), sync_loss=False, pair_offset=eng.equal_shard_pair_offset(x.shape[0]))
        # manual optimisation: Lightning counts optimizer steps, not training_step calls.  Stepping the (arithmetic-
        # free) EngineOptimizer through self.optimizers() — Lightning's wrapper — advances trainer.global_step, which
        # ModelCheckpoint, max_steps and the loggers key on.
        if _Base is not torch.nn.Module and getattr(self, "_trainer", None) is not None:
            opt = self.optimizers()
            (opt[0] if isinstance(opt, (list, tuple)) else opt).step()
        return {"loss": loss.detach().clone().reshape(())}

    @ops.on_device
    def cal_loss(self, pred_batch=None, gt_batch=None):
        """main.py:191-198 on device: MSE of the re-batched prediction — one HIP kernel, differentiable w.r.t.
        ``pred_batch`` (``_MSELoss``: the kernel emits d loss / d pred alongside the loss)."""
        gt = gt_batch['ipd'].to(pred_batch.device)
        nb, nt2, nf2, npair = gt.shape
        if pred_batch.shape[0] != nb * npair or tuple(pred_batch.shape[1:]) != (nt2, nf2):
            raise RuntimeError("cal_loss: pred %s does not match gt['ipd'] %s" % (tuple(pred_batch.shape), tuple(gt.shape)))
        return _MSELoss.apply(pred_batch.float(), gt.float())

    def configure_optimizers(self):
        """The optimizer (Adam, lr 1e-3; the ExponentialLR decay is applied by ``on_train_epoch_end``) lives in
        the HIP training engine; Lightning gets the ``EngineOptimizer`` shim so that its step counter, checkpoints
        and ``max_steps`` work (see the class).  ``fused_engine=False``: the reference's own dictionary (main.py:269-279)."""
        if not self.fused_engine:
            optimizer = torch.optim.Adam(self.arch.parameters(), lr=0.001)
            lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.8988, last_epoch=-1)
            return {'optimizer': optimizer, 'lr_scheduler': {'scheduler': lr_scheduler, 'monitor': 'valid/loss'}}
        return EngineOptimizer(self)

    def on_train_epoch_end(self):
        if getattr(self, "_train_engine", None) is not None:
            self._train_engine.lr *= 0.8988                                      # ExponentialLR gamma (main.py:272)

    def data_preprocess(self, mic_sig_batch=None, gt_batch=None, vad_batch=None, eps=1e-6, nor_flag=True):
        """main.py:200-265: [nb, ns, nch] -> [[nb*np, 4, 256, nt]] (reference layout), and — with ``gt_batch`` {'doa' [nb, nseg,
        2, ns], 'vad_sources' [nb, nseg, nvad, ns]} — the DP-IPD TARGETS gt_batch['ipd'] [nb, nseg, 512, np] (one HIP kernel,
        ``fnssl_dpipd_targets``, in place of the reference's per-batch numpy on the host), gt_batch['vad_sources'] = its mean
        over the segment's frames and gt_batch['doa'] on the device: returns [input, gt_batch] like the reference."""
        data = []
        if mic_sig_batch is not None:
            mic_sig_batch = mic_sig_batch.to(self.dev)
            data += [ops.preprocess(mic_sig_batch, self.ch_mode, eps, layout=1, normalise=bool(nor_flag))]
        if gt_batch is not None:
            from fnssl import doa as fdoa
            doa_b = gt_batch['doa'].to(self.dev).float()
            vad_b = gt_batch['vad_sources'].to(self.dev).float()
            ipd, vmean = fdoa.dpipd_targets(doa_b, vad_b, self.mic_location, self.ch_mode, 1, int(self.nfft / 2),
                                            int(self.nfft / 2) + 1, self.fre_max, self.speed, self.tar_useVAD)
            gt_batch['doa'] = doa_b
            gt_batch['ipd'] = ipd
            gt_batch['vad_sources'] = vmean
            data += [gt_batch]
        return data

    @torch.no_grad()
    @ops.on_device
    def predict_step(self, batch, batch_idx: int = 0):
        """batch [nb, nch, ns] -> preds [nb*np, nt//12, 512]  (main.py:184-189).

        Fused: STFT -> pair features land directly in the [nb', nt, nf, 4] layout the
        LSTM kernels read, so neither the reference's [nb', 4, nf, nt] tensor nor its
        permute is materialised.
        """
        sig = batch.permute(0, 2, 1).to(self.dev)
        x0 = ops.preprocess(sig, self.ch_mode, 1e-6, layout=0)
        return self.arch.forward_seq(x0)
