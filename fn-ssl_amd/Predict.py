#!/usr/bin/env python3
"""Predict entry of the MI355X path (counterpart of the reference's FN-SSL/Predict.py and of
`python main.py predict` in FN-SSL/Lightning): waveforms -> DP-IPD predictions.

Flag surface = the reference's FN-SSL/Opt.py:21-43 (`--gpu-id --workers --no-cuda --use-amp --seed --train --test
--dev --checkpoint-start --time --sources --source-state --localize-mode --bz a b c --epochs --lr --datasetMode`),
same names, types and defaults, mapped onto this path:

  --test                   the stage this entry implements (assumed when no stage flag is given)
  --gpu-id 0,1             first id = the MI355X this process drives (one process per GPU)
  --no-cuda                refused: the path has no CPU implementation (torch.cuda.* IS the HIP device on ROCm)
  --use-amp                reduced-precision matrix products = the bf16-MFMA fast mode (fp32 accumulate / tensors)
  --bz a b c               c = utterances per forward (Predict.py:76 uses args.bz[2])
  --localize-mode M K n    IPD -> DOA back end (Learner.py:222-239): method 'IDL', 'kNum' | 'unkNum', max sources
  --sources n ..           max(n) sources if --localize-mode gives none
  --datasetMode            'simulate' / 'locata' datasets (gpuRIR, LOCATA readers) are outside this path
                           (SURVEY.md 8): give the signals with --wav or --synthetic instead
  --train / --dev, --workers, --epochs, --lr, --checkpoint-start, --time, --source-state   parsed; --train / --dev exit
                           with a pointer to predict_step.MyModel.training_step (the training loop is Lightning's)

Additions of this entry: --wav, --synthetic, --nch, --seconds, --ch-mode, --checkpoint (the reference hard-codes
'/exp/04231627/', Predict.py:65; both checkpoint formats of Learner.py:318-353 load), --out, --doa-out, --mic-pos.

    python Predict.py --test --synthetic 4 --nch 4 --seconds 4.79 --bz 1 1 4 --out pred.npy
    python Predict.py --test --wav a.wav b.wav --checkpoint best_model.tar --localize-mode IDL kNum 1 --doa-out doa.npz
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


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # ---- the reference's flags (Opt.py:21-43), same names / types / defaults
    ap.add_argument('--gpu-id', type=str, default='0,1', metavar='GPU')
    ap.add_argument('--workers', type=int, default=0, metavar='Worker')
    ap.add_argument('--no-cuda', action='store_true', default=False)
    ap.add_argument('--use-amp', action='store_true', default=False)
    ap.add_argument('--seed', type=int, default=1, metavar='Seed')
    ap.add_argument('--train', action='store_true', default=False)
    ap.add_argument('--test', action='store_true', default=False)
    ap.add_argument('--dev', action='store_true', default=False)
    ap.add_argument('--checkpoint-start', action='store_true', default=False)
    ap.add_argument('--time', type=str, default='', metavar='Time')
    ap.add_argument('--sources', type=int, nargs='+', default=[1], metavar='Sources')
    ap.add_argument('--source-state', type=str, default='mobile', metavar='SourceState')
    ap.add_argument('--localize-mode', type=str, nargs='+', default=['IDL', 'kNum', 1], metavar='LocalizeMode')
    ap.add_argument('--bz', type=int, nargs='+', default=[1, 1, 1], metavar='TrainValTestBatch')
    ap.add_argument('--epochs', type=int, default=100, metavar='Epoch')
    ap.add_argument('--lr', type=float, default=0.001, metavar='LR')
    ap.add_argument('--datasetMode', type=str, default='simulate', metavar='datasetMode')
    # ---- this entry's own
    ap.add_argument("--ch-mode", default="MM", choices=["M", "MM"])
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--wav", nargs="*", default=None, help="multi-channel 16 kHz wav files")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic utterances")
    ap.add_argument("--nch", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=4.79)
    ap.add_argument("--out", default="pred.npy")
    ap.add_argument("--doa-out", default=None, help="also run the IPD->DOA back end and save DOA / VAD tracks (.npz)")
    ap.add_argument("--mic-pos", default=None, help=".npy with the [nch, 3] microphone positions "
                                                   "(default: the reference's dual-channel array, Dataset.py:87-96)")
    return ap


def parse_args(argv=None):
    """Opt.parse (Opt.py:15-51): exactly one stage; here a missing stage flag means --test."""
    args = build_parser().parse_args(argv)
    if args.train + args.test + args.dev == 0:
        args.test = True
    if args.train + args.test + args.dev != 1:
        raise Exception('Stage of train or test is unrecognized')          # Opt.py:48-49
    if len(args.bz) == 1:
        args.bz = args.bz * 3
    if len(args.bz) != 3:
        raise SystemExit("--bz takes three batch sizes (train, validation, test), got %s" % args.bz)
    lm = list(args.localize_mode) + ['IDL', 'kNum', max(args.sources)][len(args.localize_mode):]
    args.localize_mode = [lm[0], lm[1], int(lm[2])]
    return args


def main(argv=None):
    args = parse_args(argv)
    if args.train or args.dev:
        raise SystemExit("Predict.py implements the --test stage; the training step of this path is "
                         "predict_step.MyModel.training_step / fnssl.train.TrainEngine (INTEGRATION.md §3)")
    if args.no_cuda or not torch.cuda.is_available():           # Predict.py:26-27
        raise SystemExit("Predict.py: no ROCm device in use (--no-cuda / none visible); this path has no CPU implementation")
    if args.wav is None and args.synthetic <= 0:
        raise SystemExit("--datasetMode %s: the gpuRIR / LOCATA dataset readers are outside this path (SURVEY.md 8); "
                         "give the signals with --wav files or --synthetic N" % args.datasetMode)
    gpu = int(str(args.gpu_id).split(",")[0])
    dev = torch.device("cuda", gpu)
    torch.cuda.set_device(dev)                                   # every kernel launches on this device's streams
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    model = ps.MyModel(ch_mode=args.ch_mode, device=str(dev), method_mode=args.localize_mode[0],
                       source_num_mode=args.localize_mode[1], max_num_sources=args.localize_mode[2])
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint)
    model = model.to(dev).eval()
    if args.use_amp:
        model.arch.bfloat16()
    print("# Parameters:", sum(p.numel() for p in model.arch.parameters()) / 1e6, "M")

    if args.wav:
        batch = torch.from_numpy(read_wavs(args.wav))
    else:
        ns = int(args.seconds * 16000)
        batch = torch.randn(args.synthetic, args.nch, ns) * 0.05

    bz = args.bz[2]
    preds = []
    for lo in range(0, batch.shape[0], bz):
        preds.append(model.predict_step(batch[lo:lo + bz].to(dev), lo // bz).float())
    pred = torch.cat(preds)
    np.save(args.out, pred.cpu().numpy())
    print("DP-IPD predictions", tuple(pred.shape), "->", args.out)
    if args.doa_out:
        import Module as at_module
        mic = np.load(args.mic_pos) if args.mic_pos else None
        if mic is None and batch.shape[1] != 2:
            raise SystemExit("--doa-out with %d microphones needs --mic-pos" % batch.shape[1])
        get = at_module.PredDOA(method_mode=args.localize_mode[0], source_num_mode=args.localize_mode[1],
                                max_num_sources=args.localize_mode[2], ch_mode=args.ch_mode, device=str(dev),
                                mic_location=mic).to(dev)
        out, _ = get.predgt2DOA(pred_batch=pred)
        np.savez(args.doa_out, doa=out['doa'].cpu().numpy(), vad_sources=out['vad_sources'].cpu().numpy())
        print("DOA tracks", tuple(out['doa'].shape), "->", args.doa_out)
    return pred


if __name__ == "__main__":
    main()
