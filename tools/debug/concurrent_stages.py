#!/usr/bin/env python
"""Debug: which stage of enhance_batch differs when micro-batches run on several streams? (GPU)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncsnpp_ref as NR
from storm_amd import distributed as D
from storm_amd.model import ScoreModel

dev = torch.device("cuda:0")
COMMON = dict(sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15, spec_abs_exponent=0.5, nf=16)
m = ScoreModel(backbone="ncsnpp", **COMMON)
m.dnn.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(nf=16, input_channels=4), seed=5))
m.eval(no_ema=True)
m = m.to(dev)
m.set_precision("fp16")
g = torch.Generator().manual_seed(77)
lens = [int(v) for v in torch.randint(6000, 30001, (11,), generator=g)]
batches = []
for ids in D.bucket_by_frames(lens, 3):
    bl = [lens[i] for i in ids]
    yb = torch.zeros(len(ids), max(bl))
    for k, n_ in enumerate(bl):
        yb[k, :n_] = 0.1 * torch.randn(n_, generator=g)
    batches.append((yb.to(dev), None if len(set(bl)) == 1 else bl))
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
if len(sys.argv) > 2 and sys.argv[2] == "cachelen":
    from storm_amd import ops as _ops
    _cache = {}
    def _row_len_cached(lengths, like):
        if lengths is None:
            return None
        key = (tuple(int(v) for v in lengths), str(like.device))
        if key not in _cache:
            _cache[key] = torch.tensor(list(key[0]), dtype=torch.int32, device=like.device)
            torch.cuda.synchronize()
        return _cache[key]
    _ops._row_len = _row_len_cached


def one(kb):
    k, (yb, bl) = kb
    Y, peak, T_orig = m._prepare(yb, bl)
    out = {"Y": Y.clone(), "peak": peak.clone()}
    Y2, peak2, _ = m._prepare(yb, bl)
    out["Y2"] = Y2.clone(); out["peak2"] = peak2.clone(); out["yb"] = yb.clone()
    from storm_amd import ops
    win, tw = ops.dft_tables(510, yb.device, "hann")
    out["win"] = win.clone(); out["tw"] = tw.clone()
    if mode != "frontend":
        sampler = m.get_pc_sampler("reverse_diffusion", "ald", Y, N=3, corrector_steps=1, snr=0.5, intermediate=False, langevin_per_row=True, seed=100 + k)
        sample, nfe = sampler()
        out["sample"] = sample.clone()
        out["wav"] = m.data_module.spec_to_wav(sample, T_orig, peak, lengths=bl)
    return out


ref = D.run_concurrent(one, list(enumerate(batches)), 1)
ref = [{k: v.cpu() for k, v in r.items()} for r in ref]
for n in (2, 3, 3, 3, 2, 3, 2, 3, 3, 2, 3, 3):
    got = D.run_concurrent(one, list(enumerate(batches)), n)
    msg = []
    for k, (a, b) in enumerate(zip(ref, got)):
        for key in a:
            x, y = a[key], b[key].cpu()
            if not torch.equal(x, y):
                rows = [r for r in range(x.shape[0]) if not torch.equal(x[r], y[r])]
                extra = ""
                if key in ("Y", "Y2"):
                    for r in rows:
                        a_, b_ = x[r].abs().flatten(), y[r].abs().flatten()
                        sel = a_ > 1e-3
                        ratio = (b_[sel] / a_[sel])
                        extra += f" row{r}: |Y|/|ref| mean {float(ratio.mean()):.6f} std {float(ratio.std()):.2e} (ref peaks {[round(float(v), 5) for v in a['peak']]}, prev batch peaks {[round(float(v), 5) for v in ref[k - 1]['peak']] if k else None})"
                msg.append((k, key, rows, float((x - y).abs().max()), extra))
    print(f"{n} lanes:", "OK" if not msg else msg)
