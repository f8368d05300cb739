#!/usr/bin/env python3
"""Predict entry of the MI355X path (counterpart of the reference's FN-SSL/Predict.py and of
`python main.py predict` in FN-SSL/Lightning): waveforms -> DP-IPD predictions.

The reference's Predict.py wires datasets (LOCATA / simulated, gpuRIR), the IPD->DOA back
end and metrics around the forward; those are outside this path.  This entry
keeps the flags that matter for the forward (`--gpu-id`, `--bz`, `--seed`, checkpoint
loading with the reference's two checkpoint formats, Learner.py:318-353) and reads plain
.wav files or synthesises signals.

    python Predict.py --synthetic 4 --nch 4 --seconds 4.79 --out pred.npy
    python Predict.py --wav a.wav b.wav --checkpoint best_model.tar --out pred.npy
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import predict_step as ps  # noqa: E402


def load_checkpoint(model: "ps.MyModel", path: str):
    """Accepts the torch runner's {'model': state_dict} (Learner.py:351-353, optionally with the
    DataParallel 'module.' prefix) and Lightning's {'state_dict': {'arch.*': ...}} (:331-338)."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt.get("model", ckpt))
    clean = {}
    for k, v in sd.items():
        for pre in ("module.", "arch.", "_orig_mod."):
            if k.startswith(pre):
                k = k[len(pre):]
        clean[k] = v
    model.arch.load_state_dict(clean)


def read_wavs(paths):
    from scipy.io import wavfile
    sigs = []
    for p in paths:
        fs, x = wavfile.read(p)
        if fs != 16000:
            raise SystemExit("%s: expected 16 kHz audio (Predict.py:29), got %d" % (p, fs))
        x = x.astype(np.float32) / (32768.0 if x.dtype == np.int16 else 1.0)
        sigs.append(x.reshape(len(x), -1).T)                   # [nch, ns]
    ns = min(s.shape[1] for s in sigs)
    return np.stack([s[:, :ns] for s in sigs]).astype(np.float32)   # [nb, nch, ns]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpu-id", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--bz", type=int, default=32, help="utterances per forward (Opt.py: --bz)")
    ap.add_argument("--ch-mode", default="MM", choices=["M", "MM"])
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--wav", nargs="*", default=None, help="multi-channel 16 kHz wav files")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic utterances")
    ap.add_argument("--nch", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=4.79)
    ap.add_argument("--out", default="pred.npy")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("Predict.py: no ROCm device visible; this path has no CPU implementation")
    dev = torch.device("cuda", args.gpu_id)
    torch.manual_seed(args.seed)
    model = ps.MyModel(ch_mode=args.ch_mode, device=str(dev))
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint)
    model = model.to(dev).eval()
    print("# Parameters:", sum(p.numel() for p in model.arch.parameters()) / 1e6, "M")

    if args.wav:
        batch = torch.from_numpy(read_wavs(args.wav))
    elif args.synthetic > 0:
        ns = int(args.seconds * 16000)
        batch = torch.randn(args.synthetic, args.nch, ns) * 0.05
    else:
        raise SystemExit("give --wav files or --synthetic N")

    preds = []
    for lo in range(0, batch.shape[0], args.bz):
        preds.append(model.predict_step(batch[lo:lo + args.bz].to(dev), lo // args.bz).cpu())
    pred = torch.cat(preds).numpy()
    np.save(args.out, pred)
    print("DP-IPD predictions", pred.shape, "->", args.out)


if __name__ == "__main__":
    main()
