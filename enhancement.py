#!/usr/bin/env python
"""Inference CLI — drop-in for the reference enhancement.py (same flags, enhancement.py:28-36):

    python enhancement.py --test_dir noisy/ --enhanced_dir out/ --ckpt model.ckpt --mode storm \
        [--corrector ald --corrector-steps 1 --snr 0.5 --N 50]

Additions: --precision {fp32,bf16}, --batch (equal-length utterances per sampler call), --seed and multi-GPU
sharding when launched with torchrun (one process per GPU, files dealt by length; no collectives in the
sampler).  WAV I/O uses scipy.io.wavfile (torchaudio is not required)."""
import glob
import os
from argparse import ArgumentParser

import numpy as np
import torch


def read_wav(path):
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.T
    else:
        x = x[None]
    return torch.from_numpy(np.ascontiguousarray(x)), sr


def write_wav(path, x, sr):
    from scipy.io import wavfile
    wavfile.write(path, sr, x.detach().cpu().numpy().astype(np.float32))


def main():
    p = ArgumentParser()
    p.add_argument("--test_dir", type=str, required=True, help="Directory containing your corrupted files to enhance.")
    p.add_argument("--enhanced_dir", type=str, required=True, help="Where to write your cleaned files.")
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--mode", required=True, choices=["score-only", "denoiser-only", "storm"])
    p.add_argument("--corrector", type=str, choices=("ald", "langevin", "none"), default="ald")
    p.add_argument("--corrector-steps", type=int, default=1)
    p.add_argument("--snr", type=float, default=0.5)
    p.add_argument("--N", type=int, default=50)
    p.add_argument("--sampler", choices=("pc", "ode"), default="pc", help="(extension) ode: the probability-flow RK45 sampler of ScoreModel.enhance(sampler_type='ode') - "
                   "BASELINE.json configs[4]; score-only mode, one step controller per file (the reference integrates one file per solve_ivp call)")
    p.add_argument("--precision", choices=("fp32", "bf16", "fp16"), default="fp32")
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--seed", type=int, default=None, help="Philox seed of the sampler noise (default: drawn from torch's RNG, as the reference)")
    p.add_argument("--batch-invariant", action="store_true", help="every utterance's result independent - bit for bit - of what it is batched with "
                   "(storm_amd.set_batch_invariant: launch decisions per image; costs the batch-aware kernel selections)")
    p.add_argument("--group", type=int, default=8, help="score-only and storm modes: this many micro-batches (frame buckets of different lengths) run their samplers in lockstep "
                   "and share the launches of the score network (ScoreModel.enhance_stream); 1 = one micro-batch after the other")
    p.add_argument("--dist-world1", action="store_true", help="with ONE rank: form the RCCL process group anyway (dry run of the sharded path on one GPU)")
    args = p.parse_args()
    if args.sampler == "ode" and args.mode != "score-only":
        raise SystemExit("--sampler ode: score-only mode (the reference's StoRM ODE path drops the conditioning, model.py:671-691)")

    from storm_amd import distributed as D
    from storm_amd.model import DiscriminativeModel, ScoreModel, StochasticRegenerationModel
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    rank, world, local = D.init(single_rank_group=args.dist_world1)
    if world > 1:
        D.pin_to_gpu_numa(local)                            # this rank's launch thread next to its GPU
    os.makedirs(args.enhanced_dir, exist_ok=True)
    model_cls = {"storm": StochasticRegenerationModel, "score-only": ScoreModel, "denoiser-only": DiscriminativeModel}[args.mode]
    model = model_cls.load_from_checkpoint(args.ckpt, base_dir="", batch_size=1, num_workers=0, kwargs=dict(gpu=False))
    model.eval(no_ema=False)
    model.cuda()
    model.set_precision(args.precision)
    if args.batch_invariant:
        import storm_amd
        storm_amd.set_batch_invariant(True)

    files = sorted(glob.glob(os.path.join(args.test_dir, "*.wav")))
    wavs, lengths = [], []
    for f in files:
        y, sr = read_wav(f)
        assert sr == 16000, "You need to make sure sample_sr matches model_sr --> resample to 16kHz"
        wavs.append(y[:1])
        lengths.append(y.shape[1])
    mine = D.shard_indices(len(files), rank, world, lengths)
    # micro-batches of utterances that share a padded frame count (different lengths welcome): equal to per-file runs
    def run(batch):
        ids = [mine[k] for k in batch]
        lens = [lengths[i] for i in ids]
        y = torch.zeros(len(ids), max(lens))
        for k, i in enumerate(ids):
            y[k, :lens[k]] = wavs[i][0]
        ragged = None if len(set(lens)) == 1 else lens
        if args.mode == "denoiser-only":
            outs = [model.enhance(wavs[i]) for i in ids]
        else:
            kw = {} if args.seed is None else dict(seed=args.seed + ids[0])     # distinct, reproducible draws per batch
            x_hat = model.enhance_batch(y, lengths=ragged, **skw, **kw)
            outs = [x_hat[k, :lens[k]] for k in range(len(ids))]
        return ids, outs

    skw = dict(sampler_type="ode", N=args.N) if args.sampler == "ode" else dict(corrector=args.corrector, N=args.N, corrector_steps=args.corrector_steps, snr=args.snr)
    buckets = D.bucket_by_frames([lengths[i] for i in mine], args.batch)
    if args.mode in ("score-only", "storm") and args.group > 1 and len(buckets) > 1:
        # a ragged set of files: micro-batches of 2 - 3 rows each - their score evaluations share launches (storm_ncsnpp_forward_group)
        # (--group micro-batches in flight; one that finishes - an ODE micro-batch needs its own number of evaluations - is replaced by the next)
        chunk, metas = [], []
        for batch in buckets:
            ids = [mine[k] for k in batch]
            lens = [lengths[i] for i in ids]
            y = torch.zeros(len(ids), max(lens))
            for k, i in enumerate(ids):
                y[k, :lens[k]] = wavs[i][0]
            chunk.append((y, None if len(set(lens)) == 1 else lens))
            metas.append((ids, lens))
        outs = model.enhance_stream(chunk, width=args.group, seeds=None if args.seed is None else [args.seed + ids[0] for ids, _ in metas],     # (the draws of the one-by-one path)
                                    **skw)
        for (ids, lens), x_hat in zip(metas, outs):
            for k, i in enumerate(ids):
                write_wav(os.path.join(args.enhanced_dir, os.path.basename(files[i])), x_hat[k, :lens[k]].float().reshape(-1), 16000)
        buckets = []
    for batch in buckets:
        ids, outs = run(batch)
        for i, x in zip(ids, outs):
            write_wav(os.path.join(args.enhanced_dir, os.path.basename(files[i])), x.float().reshape(-1), 16000)
    D.finish()


if __name__ == "__main__":
    main()
