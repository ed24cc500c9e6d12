#!/usr/bin/env python
"""Race screen at the bench shape: the full 27.8 M-parameter score network (batch 16, 256 x 512, bf16) is evaluated
repeatedly on the same input; any timing-dependent hazard in the LDS-DMA / counted-vmcnt pipelines shows up as a
bitwise difference between repetitions.  Also repeats a seeded 2-step sampler run."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import ncsnpp_ref as NR  # noqa: E402  (test infrastructure: seeded weights only)
from storm_amd.model import ScoreModel  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
large = len(sys.argv) > 2 and sys.argv[2] == "large"     # ncsnpplarge at configs[3]'s shape: the split-K launch pairs of its deep levels
dev = torch.device("cuda:0")
if large:
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15, spec_abs_exponent=0.5)
    m.dnn.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(**NR.NAMED_CONFIGS["ncsnpplarge"]), seed=7))
    m._error_loading_ema = True
    m = m.eval().to(dev)
    m.set_precision("bf16")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 1, 256, 1024, dtype=torch.complex64, generator=g).to(dev)
    y = torch.randn(8, 1, 256, 1024, dtype=torch.complex64, generator=g).to(dev)
    t = torch.linspace(0.1, 0.9, 8).to(dev)
    bad = 0
    with torch.no_grad():
        ref = m(x, t, y).clone()
        for i in range(reps):
            if not torch.equal(m(x, t, y), ref):
                bad += 1
                print("forward repetition", i, "differs")
    print("ncsnpplarge 8 x 256 x 1024: finite:", bool(torch.isfinite(ref.abs()).all()), " RESULT", "FAIL" if bad else "PASS")
    sys.exit(1 if bad else 0)
if len(sys.argv) > 2 and sys.argv[2] == "invariant":     # the batch-invariant mode at the bench shape: every row alone / in 2s / 4s / 8s == the row in the batch of 16
    import storm_amd
    m = ScoreModel(backbone="ncsnpp", sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15, spec_abs_exponent=0.5)
    m.dnn.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(), seed=7))
    m._error_loading_ema = True
    m = m.eval().to(dev)
    bad = n = 0
    storm_amd.set_batch_invariant(True)
    with torch.no_grad():
        for prec in ("bf16", "fp16"):
            m.set_precision(prec)
            for seed in range(reps):
                g = torch.Generator().manual_seed(100 + seed)
                x = torch.randn(16, 1, 256, 512, dtype=torch.complex64, generator=g).to(dev)
                y = torch.randn(16, 1, 256, 512, dtype=torch.complex64, generator=g).to(dev)
                t = (0.03 + 0.97 * torch.rand(16, generator=g)).to(dev)
                ref = m(x, t, y).clone()
                for k in (1, 2, 4, 8):
                    for b0 in range(0, 16, k):
                        n += 1
                        if not torch.equal(m(x[b0:b0 + k], t[b0:b0 + k], y[b0:b0 + k]), ref[b0:b0 + k]):
                            bad += 1
                            print(prec, "seed", seed, "rows", b0, "...", b0 + k - 1, "differ from the batch of 16")
    storm_amd.set_batch_invariant(False)
    print(f"batch-invariant mode, 16 x 256 x 512, bf16 + fp16, {reps} inputs each: {n} sub-batch evaluations (1 / 2 / 4 / 8 rows) against the batch of 16:", bad, "differ.  RESULT", "FAIL" if bad else "PASS")
    sys.exit(1 if bad else 0)
small = len(sys.argv) > 2 and sys.argv[2] == "small"     # one / two / four utterances per call: the small-call split-K pairs and the attention key split (round 5)
m = ScoreModel(backbone="ncsnpp", sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15, spec_abs_exponent=0.5)
m.dnn.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(), seed=7))
m._error_loading_ema = True
m = m.eval().to(dev)
m.set_precision("bf16")
g = torch.Generator().manual_seed(0)
if small:
    bad = 0
    with torch.no_grad():
        for B in (1, 2, 4):
            x = torch.randn(B, 1, 256, 512, dtype=torch.complex64, generator=g).to(dev)
            y = torch.randn(B, 1, 256, 512, dtype=torch.complex64, generator=g).to(dev)
            t = torch.linspace(0.1, 0.9, B).to(dev)
            ref = m(x, t, y).clone()
            for i in range(reps):
                if not torch.equal(m(x, t, y), ref):
                    bad += 1
                    print(f"batch {B}: forward repetition {i} differs")
            wav = (0.1 * torch.randn(B, 64000, generator=g)).to(dev)
            w0 = m.enhance_batch(wav, N=3, corrector="ald", snr=0.5, seed=11).clone()
            for i in range(3):
                if not torch.equal(w0, m.enhance_batch(wav, N=3, corrector="ald", snr=0.5, seed=11)):
                    bad += 1
                    print(f"batch {B}: sampler repetition {i} differs")
            print(f"ncsnpp {B} x 256 x 512: finite {bool(torch.isfinite(ref.abs()).all())}")
    print("small calls (1 / 2 / 4 utterances): RESULT", "FAIL" if bad else "PASS")
    sys.exit(1 if bad else 0)
x = torch.randn(16, 1, 256, 512, dtype=torch.complex64, generator=g).to(dev)
y = torch.randn(16, 1, 256, 512, dtype=torch.complex64, generator=g).to(dev)
t = torch.linspace(0.1, 0.9, 16).to(dev)
bad = 0
with torch.no_grad():
    ref = m(x, t, y).clone()
    for i in range(reps):
        out = m(x, t, y)
        if not torch.equal(out, ref):
            bad += 1
            print("forward repetition", i, "differs: max", float((out - ref).abs().max()))
    wav = (0.1 * torch.randn(16, 32000, generator=g)).to(dev)
    w0 = m.enhance_batch(wav, N=2, corrector="ald", snr=0.5, seed=11).clone()
    for i in range(max(2, reps // 5)):
        w1 = m.enhance_batch(wav, N=2, corrector="ald", snr=0.5, seed=11)
        if not torch.equal(w0, w1):
            bad += 1
            print("sampler repetition", i, "differs: max", float((w0 - w1).abs().max()))
print("finite:", bool(torch.isfinite(ref.abs()).all()), " RESULT", "FAIL" if bad else "PASS")
sys.exit(1 if bad else 0)
