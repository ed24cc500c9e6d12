"""Evaluation loop — drop-in for sgmse/util/inference.py:20-72 (`evaluate_model`), built for the batched engine.

The reference enhances the first `num_eval_files` validation pairs ONE BY ONE (model.enhance) and averages PESQ,
SI-SDR and ESTOI on the host.  Here the pairs are grouped into micro-batches of utterances that share a padded frame
count (storm_amd.distributed.bucket_by_frames: different lengths welcome; every op of the path is per utterance, so a
batched run equals the per-file runs), enhanced with `model.enhance_batch(lengths=...)`, SI-SDR is one HIP launch per batch
(storm_si_sdr), and PESQ / ESTOI are used when the `pesq` / `pystoi` packages are importable (they are CPU reference
implementations of ITU-T P.862 / ESTOI; without them the two averages are NaN).

Pairs come from `model.data_module.valid_set.__getitem__(i, raw=True)` like the reference, or from `pairs=` (a sequence
of (clean [1, L], noisy [1, L]) tensors) since datasets are outside the hot path."""
import math

import torch

from ..distributed import bucket_by_frames
from .other import si_sdr_batch

# Settings of the reference's validation runs (util/inference.py:11-13)
snr = 0.5
N = 50
corrector_steps = 1
MAX_VIS_SAMPLES = 10


def _optional(name, attr):
    try:
        return getattr(__import__(name), attr)
    except Exception:
        return None


def evaluate_model(model, num_eval_files, spec=False, audio=False, discriminative=False, pairs=None, batch=16, noise_for=None,
                   **enhance_kwargs):
    """Returns (pesq, si_sdr, estoi, [noisy, estimate, clean spectrograms] | None, [noisy, estimate, clean audio] | None),
    the reference's tuple.  `discriminative` is accepted for signature parity (the model class decides the path).
    noise_for(ids) -> noise_fn: injected sampler noise for the micro-batch of files `ids` (parity tests; production runs
    pass seed= and draw in-kernel)."""
    model.eval()
    pesq, stoi = _optional("pesq", "pesq"), _optional("pystoi", "stoi")
    if pairs is None:
        vs = model.data_module.valid_set
        pairs = [vs.__getitem__(i, raw=True) for i in range(num_eval_files)]
    pairs = [(x if x.dim() == 2 else x.unsqueeze(0), y if y.dim() == 2 else y.unsqueeze(0)) for x, y in list(pairs)[:num_eval_files]]
    n = len(pairs)
    dev = next(model.parameters()).device
    est = [None] * n
    sdr = torch.zeros(n, dtype=torch.float64)
    hop = model.data_module.hop_length
    batched = hasattr(model, "enhance_batch") and not discriminative
    for ids in bucket_by_frames([p[1].shape[-1] for p in pairs], batch if batched else 1, hop=hop):
        lens = [pairs[i][1].shape[-1] for i in ids]
        width = max(lens)
        y = torch.zeros(len(ids), width)
        x = torch.zeros(len(ids), width)
        for k, i in enumerate(ids):                                         # first channel only (util/inference.py:44-47)
            y[k, :lens[k]] = pairs[i][1][0]
            x[k, :min(lens[k], pairs[i][0].shape[-1])] = pairs[i][0][0, :lens[k]]
        y, x = y.to(dev), x.to(dev)
        kw = dict(enhance_kwargs)
        if noise_for is not None:
            kw["noise_fn"] = noise_for(ids)
        if batched:
            x_hat = model.enhance_batch(y, lengths=lens if len(set(lens)) > 1 else None, **kw)
        else:
            x_hat = model.enhance(y[:1], **kw).reshape(1, -1)         # (N / snr / corrector_steps ... reach the model here too: util/inference.py:49)
        x_hat = x_hat.reshape(len(ids), -1).float().to(dev)
        if len(set(lens)) == 1:                                             # one launch for the whole micro-batch
            sdr[ids] = si_sdr_batch(x.contiguous(), x_hat.contiguous()).double().cpu()
        else:                                                               # ragged rows: every file over its own length
            for k, i in enumerate(ids):
                sdr[i] = float(si_sdr_batch(x[k:k + 1, :lens[k]], x_hat[k:k + 1, :lens[k]]))
        for k, i in enumerate(ids):
            est[i] = x_hat[k, :lens[k]].cpu()
    _pesq = _estoi = float("nan")
    if pesq is not None:
        _pesq = sum(pesq(16000, pairs[i][0][0].numpy(), est[i].numpy(), "wb") for i in range(n)) / n
    if stoi is not None:
        _estoi = sum(stoi(pairs[i][0][0].numpy(), est[i].numpy(), 16000, extended=True) for i in range(n)) / n
    specs = audios = None
    if spec:
        k = min(n, MAX_VIS_SAMPLES)
        sp = lambda w: model._stft(w.to(dev)).cpu()
        specs = [[sp(pairs[i][1][0]) for i in range(k)], [sp(est[i]) for i in range(k)], [sp(pairs[i][0][0]) for i in range(k)]]
    if audio:
        k = min(n, MAX_VIS_SAMPLES)
        audios = [[pairs[i][1][0] for i in range(k)], [est[i] for i in range(k)], [pairs[i][0][0] for i in range(k)]]
    return _pesq, float(sdr.mean()) if n else math.nan, _estoi, specs, audios
