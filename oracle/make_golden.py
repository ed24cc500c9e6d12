"""Generate tests/golden/*.npz by running the REFERENCE (imported on CPU from
/root/reference with the stub recipe in oracle/ref_import.py) on seeded inputs,
and check the oracle restatement against it while doing so.

Container-only: run ``python -m oracle.make_golden`` from the repo root.  The
fixtures are data (inputs + the reference's outputs); no reference source is
stored.  Weights are regenerated on both sides from ``seeded_state_dict(cfg, seed)``
(a SHA-256 of the weight blob is stored to detect RNG drift).
"""
import hashlib
import math
import os
import sys
import types

import numpy as np
import torch

from oracle import ncsnpp_ref as NR
from oracle import sde_ref as SR
from oracle import frontend_ref as FR
from oracle.ref_import import import_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TINY = {
    # name: (cfg kwargs, spatial shape)
    "tiny4": (dict(nf=8, input_channels=4), (32, 64)),
    "tiny6": (dict(nf=8, input_channels=6), (32, 64)),
    "tiny2d": (dict(nf=8, input_channels=2, discriminative=True), (32, 64)),
    "tinyattn": (dict(nf=16, ch_mult=(1, 1, 2, 2), num_res_blocks=2, attn_resolutions=(8,),
                      image_size=32, input_channels=4), (32, 64)),
}


# F18: OUVPSDE parameter sets (beta_min, beta_max, stiffness) and sampler settings (N, predictor, corrector, corrector steps)
OUVP_CASES = {"a": (0.1, 2.0, 1), "b": (0.05, 1.2, 2.0)}
OUVP_SAMPLERS = {"lang": (5, "reverse_diffusion", "langevin", 1), "em": (6, "euler_maruyama", "none", 1),
                 "none": (4, "reverse_diffusion", "none", 1), "lang2": (3, "euler_maruyama", "langevin", 2)}


# F20: data-module settings beside the defaults (window, n_fft, hop_length, spec_factor, spec_abs_exponent)
F20_CASES = {"sqrthann": dict(window="sqrthann"), "lin": dict(spec_abs_exponent=1.0, spec_factor=0.33),
             "sq667": dict(window="sqrthann", spec_abs_exponent=0.667, spec_factor=0.065), "n254": dict(n_fft=254, hop_length=64),
             "hop256": dict(hop_length=256)}

# F19: shapes (B, F, T) of the small variants' forwards; inputs regenerate from a seed on both sides (their SHA-256 is stored)
F19_SHAPES = dict(small=(1, 32, 64), real=(2, 256, 128))


def f19_inputs(name, tag):
    B, F, T = F19_SHAPES[tag]
    g = torch.Generator().manual_seed(1919 + 7 * len(name) + B)
    x = torch.randn(B, 2, F, T, dtype=torch.complex64, generator=g)
    return x, torch.rand(B, generator=g) * 0.9 + 0.05


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def rel_l2(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().pow(2).sum().sqrt() / (b.abs().pow(2).sum().sqrt() + 1e-30))


def c2np(x):
    return x.detach().numpy()


def ref_backbone(ref, cfg: NR.NCSNppConfig, sd):
    kw = dict(nf=cfg.nf, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks,
              attn_resolutions=cfg.attn_resolutions, image_size=cfg.image_size,
              input_channels=cfg.input_channels, discriminative=cfg.discriminative)
    net = ref["ncsnpp"].NCSNpp(**kw)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    net.eval()
    return net


def check(name, got, want, tol):
    e = rel_l2(got, want)
    status = "ok" if e <= tol else "FAIL"
    print(f"  [{status}] oracle vs reference  {name}: rel-L2 {e:.3e} (tol {tol:g})")
    if e > tol:
        raise SystemExit(f"oracle disagrees with the reference on {name}")
    return e


def gen_f7(ref):
    """F7: probability-flow ODE sampler (scipy RK45 on the host, sampling/__init__.py:71-141) with an
    analytic score on a toy state: end point + nfev."""
    print("F7 ODE sampler")
    g = torch.Generator().manual_seed(77)
    y = torch.randn(2, 1, 8, 16, dtype=torch.complex64, generator=g) * 0.3
    z = SR.complex_randn(y.shape, g)
    sde = ref["sdes"].OUVESDE(theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30)

    def score(x, t, yy):
        return -(x - yy) / (sde._std(t)[:, None, None, None] ** 2 + 0.1)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: z.to(x.dtype)
    try:
        sampler = ref["sampling"].get_ode_sampler(sde, score, y=y, eps=0.03, device="cpu")
        x, nfe = sampler()
    finally:
        torch.randn_like = orig
    print(f"  ode: nfev {nfe}")
    np.savez_compressed(os.path.join(OUT, "f7_ode.npz"), y=c2np(y), z=c2np(z), out=c2np(x), nfe=np.array(nfe))


def seeded_input(shape, seed, scale, dtype=torch.complex64):
    """Large fixture inputs are regenerated from a seed on both sides (their SHA-256 is stored)."""
    return torch.randn(*shape, dtype=dtype, generator=torch.Generator().manual_seed(seed)) * scale


def tensor_hash(x):
    return hashlib.sha256(x.detach().contiguous().numpy().tobytes()).hexdigest()


def gen_f8(ref):
    """F8: the BENCH shape.  (a) `ncsnpp` (27.8 M, 4 input channels) forward of ONE 4-s utterance, [1,2,256,512]
    (ncsnpp.py:281-450) - the shape at which the production kernel selection (pipelined 256-cout conv, L = 2048
    attention) is active; (b) AttnBlockpp with 256 channels at 32 x 64 = 2048 positions (layerspp.py:60-91)."""
    print("F8 bench-shape forward + L=2048 attention (about a minute)")
    torch.set_num_threads(8)
    f8 = {}
    cfg = NR.NCSNppConfig(input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=11)
    net = ref_backbone(ref, cfg, sd)
    xin = seeded_input((1, 2, 256, 512), 808, 0.5)
    tt = torch.tensor([0.37])
    with torch.no_grad():
        y_ref = net(xin, tt)
        y_or = NR.ncsnpp_forward(sd, cfg, xin, tt)
    check("ncsnpp4 forward @ 256x512", y_or, y_ref, 5e-5)
    f8.update(full4_y=c2np(y_ref), full4_xhash=np.array(tensor_hash(xin)), full4_sdhash=np.array(sd_hash(sd)),
              t=np.array([0.37], dtype=np.float32))
    C = 256
    blk = ref["layerspp"].AttnBlockpp(channels=C, skip_rescale=True, init_scale=0.)
    ga = torch.Generator().manual_seed(809)
    sda = {k: (torch.randn(v.shape, generator=ga) * ((1.5 / C ** 0.5) if "NIN" in k and k.endswith("W") else 0.1)
               + (1.0 if "GroupNorm_0.weight" in k else 0.0)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sda)
    xa = seeded_input((1, C, 32, 64), 810, 1.0, torch.float32)
    with torch.no_grad():
        ya = blk(xa)
    check("attnblock L=2048", NR.attnblock(NR._SD(sda), xa), ya, 1e-5)
    f8.update(attn_y=ya.numpy(), attn_xhash=np.array(tensor_hash(xa)), **{"attn_" + k: v.numpy() for k, v in sda.items()})
    np.savez_compressed(os.path.join(OUT, "f8_bench_shape.npz"), **f8)


def gen_f9(ref):
    """F9: the remaining one-call surfaces of model.py - DiscriminativeModel.enhance (model.py:351-370) and
    StochasticRegenerationModel.enhance with denoiser_only / return_stft (model.py:720-780) - on an 8000-sample input."""
    print("F9 DiscriminativeModel.enhance, StoRM denoiser_only / return_stft")
    f9 = {}
    M = ref["model"]
    DM = ref["data_module"].SpecsDataModule
    ywav = torch.randn(1, 8000, generator=torch.Generator().manual_seed(909)) * 0.1
    common = dict(sde="ouve", data_module_cls=DM, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                  spec_factor=0.15, spec_abs_exponent=0.5, nf=8)
    m = M.DiscriminativeModel(backbone="ncsnpp", input_channels=2, discriminative=True, **common)
    cfg_d = NR.NCSNppConfig(nf=8, input_channels=2, discriminative=True)
    sd_d = NR.seeded_state_dict(cfg_d, seed=41)
    m.dnn.load_state_dict(sd_d)
    m.eval(no_ema=True)
    with torch.no_grad():
        xd_ref = m.enhance(ywav.clone())
    Y, nfac, T0 = FR.wav_to_spec(ywav)
    with torch.no_grad():
        xd_or = FR.spec_to_wav(NR.ncsnpp_forward(sd_d, cfg_d, Y, None), nfac, T0)
    check("DiscriminativeModel.enhance", xd_or, xd_ref, 1e-4)
    f9.update(wav_in=ywav.numpy(), disc_out=xd_ref.numpy())
    g = torch.Generator().manual_seed(910)
    m = M.StochasticRegenerationModel(backbone_denoiser="ncsnpp", backbone_score="ncsnpp", condition="both", **dict(common))
    cfg_s = NR.NCSNppConfig(nf=8, input_channels=6)
    sd_d2, sd_s = NR.seeded_state_dict(cfg_d, seed=42), NR.seeded_state_dict(cfg_s, seed=43)
    m.denoiser_net.load_state_dict(sd_d2)
    m.score_net.load_state_dict(sd_s)
    m.eval(no_ema=True)
    with torch.no_grad():
        x_do = m.enhance(ywav.clone(), denoiser_only=True)
    f9.update(storm_denoiser_only=x_do.numpy())
    N = 2
    noises = [SR.complex_randn(Y.shape, g) for _ in range(1 + N)]
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
    try:
        with torch.no_grad():
            sample, Yr, T_orig, nf_ = m.enhance(ywav.clone(), N=N, corrector="none", snr=0.5, return_stft=True)
    finally:
        torch.randn_like = orig
    assert T_orig == 8000 and torch.equal(Yr, Y.squeeze())
    f9.update(stft_noise=np.stack([c2np(n) for n in noises]), stft_sample=c2np(sample), stft_Y=c2np(Yr), stft_norm=np.array(nf_))
    np.savez_compressed(os.path.join(OUT, "f9_surfaces.npz"), **f9)



def gen_f10(ref):
    """F10: the evaluation harness.  (a) si_sdr / si_sdr_torch (util/other.py:82-94) on seeded (clean, estimate) pairs of
    several lengths and qualities; (b) evaluate_model (util/inference.py:20-72) on a tiny ScoreModel over three validation
    pairs of DIFFERENT lengths with `pesq` / `stoi` (absent third-party packages) stubbed to the constants 2.5 / 0.75: the
    reference's returned tuple (incl. its spec / audio lists) and the noise it drew per file."""
    print("F10 si_sdr / si_sdr_torch / evaluate_model")
    f10 = {}
    O, INF = ref["other"], ref["inference"]
    g = torch.Generator().manual_seed(1010)
    pairs = []
    for k, (n, noise) in enumerate(((16000, 0.05), (8000, 0.5), (12345, 2.0), (64000, 0.2), (1000, 1e-4))):
        s = torch.randn(n, generator=g) * 0.1
        s_hat = (0.7 + 0.1 * k) * s + noise * 0.1 * torch.randn(n, generator=g)
        sd_np = float(O.si_sdr(s.numpy(), s_hat.numpy()))
        sd_t = float(O.si_sdr_torch(s, s_hat))
        f10.update({f"pair{k}_s": s.numpy(), f"pair{k}_shat": s_hat.numpy(), f"pair{k}_si_sdr": np.array(sd_np),
                    f"pair{k}_si_sdr_torch": np.array(sd_t)})
        pairs.append((sd_np, sd_t))
    # unequal lengths: si_sdr_torch truncates to the shorter one (util/other.py:89-90)
    s, s_hat = torch.randn(5000, generator=g), torch.randn(4000, generator=g)
    s_hat = s_hat + s[:4000]
    f10.update(trunc_s=s.numpy(), trunc_shat=s_hat.numpy(), trunc_si_sdr_torch=np.array(float(O.si_sdr_torch(s, s_hat))))
    print("  si_sdr (numpy, torch):", [(round(a, 3), round(b, 3)) for a, b in pairs])

    M, DM = ref["model"], ref["data_module"].SpecsDataModule
    common = dict(sde="ouve", data_module_cls=DM, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                  spec_factor=0.15, spec_abs_exponent=0.5, nf=8)
    m = M.ScoreModel(backbone="ncsnpp", **common)
    cfg = NR.NCSNppConfig(nf=8, input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=51)
    m.dnn.load_state_dict(sd)
    lengths = (8000, 7000, 4000)                       # 63 / 55 frames -> one 64-frame bucket; 32 frames -> its own
    clean = [torch.randn(1, n, generator=g) * 0.1 for n in lengths]
    noisy = [c + 0.05 * torch.randn(c.shape, generator=g) for c in clean]

    class ValidSet:
        def __getitem__(self, i, raw=False):
            assert raw
            return clean[i].clone(), noisy[i].clone()
    m.data_module.valid_set = ValidSet()
    INF.pesq = lambda fs, x, x_hat, mode: 2.5
    INF.stoi = lambda x, x_hat, fs, extended=True: 0.75
    # the reference's loop calls model.enhance(y) with its defaults (pc, reverse_diffusion + ald, N = 50): keep the defaults but
    # a short chain, through the one knob the signature offers (the model's own enhance is called positionally with y only)
    N = 3
    orig_enh = m.enhance
    m.enhance = lambda y: orig_enh(y, N=N)
    noises = []
    orig = torch.randn_like

    def draw(x, *a, **k):
        z = SR.complex_randn(x.shape, g)
        noises.append(z)
        return z.to(x.dtype)
    torch.randn_like = draw
    try:
        with torch.no_grad():
            pq, sdr, est, specs, audios = INF.evaluate_model(m, len(lengths), spec=True, audio=True)
    finally:
        torch.randn_like = orig
    per_file = 1 + N * 2                                # prior + N x (ald corrector, predictor)
    assert len(noises) == per_file * len(lengths)
    print(f"  evaluate_model: pesq {pq} si_sdr {sdr:.4f} estoi {est}")
    f10.update(eval_pesq=np.array(pq), eval_si_sdr=np.array(sdr), eval_estoi=np.array(est), eval_N=np.array(N),
               eval_sdhash=np.array(sd_hash(sd)))
    for i in range(len(lengths)):
        f10.update({f"eval_clean{i}": clean[i].numpy(), f"eval_noisy{i}": noisy[i].numpy(),
                    f"eval_estimate{i}": audios[1][i].numpy(),
                    f"eval_noise{i}": np.stack([c2np(z) for z in noises[i * per_file:(i + 1) * per_file]]),
                    f"eval_spec_est{i}": c2np(specs[1][i])})
    np.savez_compressed(os.path.join(OUT, "f10_eval.npz"), **f10)


def gen_f11(ref):
    """F11: BASELINE.json configs[3]'s network at its own shape - `ncsnpplarge` (65.6 M, ncsnpp.py:460-470) forward of ONE 8-s
    utterance, [1,2,256,1024]: three attention blocks at 16 x 64 = 1024 positions outside the bottleneck (ncsnpp.py:338,385)."""
    print("F11 ncsnpplarge forward @ 256x1024 (a few minutes)")
    torch.set_num_threads(8)
    cfg = NR.NCSNppConfig(**NR.NAMED_CONFIGS["ncsnpplarge"], input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=11)
    net = ref_backbone(ref, cfg, sd)
    xin = seeded_input((1, 2, 256, 1024), 1111, 0.5)
    tt = torch.tensor([0.37])
    with torch.no_grad():
        y_ref = net(xin, tt)
        y_or = NR.ncsnpp_forward(sd, cfg, xin, tt)
    check("ncsnpplarge forward @ 256x1024", y_or, y_ref, 5e-5)
    np.savez_compressed(os.path.join(OUT, "f11_large_shape.npz"), large4_y=c2np(y_ref), large4_xhash=np.array(tensor_hash(xin)),
                        large4_sdhash=np.array(sd_hash(sd)), t=np.array([0.37], dtype=np.float32))


def gen_f12(ref):
    """F12: BASELINE.json configs[4] semantics - the ODE sampler as the reference MODEL runs it, one utterance per solve_ivp
    call (model.py:224-244, minibatch = 1; sampling/__init__.py:71-141).  (a) analytic score, three utterances of one toy
    shape solved one by one: end points + nfev each; (b) ScoreModel.enhance(y, sampler_type="ode") wav -> wav with a tiny
    NCSN++ on three utterances of different lengths that share the 64-frame bucket, one by one."""
    print("F12 per-utterance ODE runs")
    f12 = {}
    g = torch.Generator().manual_seed(1212)
    sde = ref["sdes"].OUVESDE(theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30)

    def score(x, t, yy):                               # stiffness grows with the utterance's level: different step sequences
        return -(x - yy) * (1 + 4 * yy.abs().mean(dim=(1, 2, 3), keepdim=True)) / (sde._std(t)[:, None, None, None] ** 2 + 0.1)
    y = torch.randn(3, 1, 8, 16, dtype=torch.complex64, generator=g) * torch.tensor([0.1, 0.4, 1.5])[:, None, None, None]
    z = SR.complex_randn(y.shape, g)
    outs, nfes = [], []
    orig = torch.randn_like
    for b in range(3):
        torch.randn_like = lambda x, *a, **k: z[b:b + 1].to(x.dtype)
        try:
            xb, nb = ref["sampling"].get_ode_sampler(sde, score, y=y[b:b + 1], eps=0.03, device="cpu")()
        finally:
            torch.randn_like = orig
        outs.append(xb); nfes.append(nb)
    print("  analytic: nfev per utterance", nfes)
    f12.update(toy_y=c2np(y), toy_z=c2np(z), toy_out=c2np(torch.cat(outs, 0)), toy_nfe=np.array(nfes))

    M, DM = ref["model"], ref["data_module"].SpecsDataModule
    common = dict(sde="ouve", data_module_cls=DM, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                  spec_factor=0.15, spec_abs_exponent=0.5, nf=8)
    m = M.ScoreModel(backbone="ncsnpp", **common)
    cfg = NR.NCSNppConfig(nf=8, input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=61)
    m.dnn.load_state_dict(sd)
    m.eval(no_ema=True)
    lengths = (8000, 7300, 6600)                       # 63 / 58 / 52 frames: all pad to 64
    for i, n in enumerate(lengths):
        wav = torch.randn(1, n, generator=g) * 0.1
        zi = SR.complex_randn((1, 1, 256, 64), g)
        torch.randn_like = lambda x, *a, **k: zi.to(x.dtype)
        try:
            with torch.no_grad():
                xh, nfe, _ = m.enhance(wav.clone(), sampler_type="ode", timeit=True, device="cpu")
        finally:
            torch.randn_like = orig
        print(f"  enhance(ode) {n} samples: nfev {nfe}")
        f12.update({f"ode_wav{i}": wav.numpy(), f"ode_z{i}": c2np(zi), f"ode_out{i}": xh.numpy(), f"ode_nfe{i}": np.array(int(nfe[0]) if isinstance(nfe, (list, tuple)) else int(nfe))})
    f12["ode_sdhash"] = np.array(sd_hash(sd))
    np.savez_compressed(os.path.join(OUT, "f12_ode_rows.npz"), **f12)


def _full_sampler_fixture(ref, tag, fname, nsamples, seeds, what, backbone="ncsnpp", N=30):
    """ScoreModel.enhance (model.py:273-310) of the seeded 27.8 M `ncsnpp` (or `backbone`) on ONE utterance of `nsamples` samples: N = 30 (or N) reverse steps,
    `reverse_diffusion` + `ald` x 1 = 2 N score evaluations inside pc_sampler (sampling/__init__.py:54-66), with RECORDED noise (61 draws,
    regenerated on both sides from a seed; their SHA-256 is stored).  Stores the reference's wav and the sampler's final spectrogram (the
    tensor enhance hands to to_audio), and checks the oracle restatement against both on the way."""
    print(f"{tag} full-width {2 * N}-evaluation enhance, {what}")
    torch.set_num_threads(16)
    M, DM = ref["model"], ref["data_module"].SpecsDataModule
    steps = 1
    seed_w, seed_n, seed_wav = seeds
    cfg = NR.NCSNppConfig(**NR.NAMED_CONFIGS[backbone], input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=seed_w)
    m = M.ScoreModel(backbone=backbone, sde="ouve", data_module_cls=DM, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                     spec_factor=0.15, spec_abs_exponent=0.5)
    m.dnn.load_state_dict(sd)
    m.eval(no_ema=True)
    ywav = torch.randn(1, nsamples, generator=torch.Generator().manual_seed(seed_wav)) * 0.1
    Ysh = FR.wav_to_spec(ywav)[0].shape
    gn = torch.Generator().manual_seed(seed_n)
    noises = [SR.complex_randn(Ysh, gn) for _ in range(1 + N * (steps + 1))]
    nhash = hashlib.sha256(b"".join(c2np(n).tobytes() for n in noises)).hexdigest()
    it = iter(noises)
    orig = torch.randn_like
    final = {}
    torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
    to_audio_orig = m.to_audio

    def to_audio_spy(spec, length=None):                  # the sampler's final state, as enhance() hands it to the back end
        final["spec"] = spec.detach().clone()
        return to_audio_orig(spec, length)
    m.to_audio = to_audio_spy
    try:
        with torch.no_grad():
            xh_ref = m.enhance(ywav.clone(), N=N, corrector="ald", corrector_steps=steps, snr=0.5)
    finally:
        torch.randn_like = orig
        m.to_audio = to_audio_orig
    Y, nfac, T0 = FR.wav_to_spec(ywav)
    it = iter(noises)
    with torch.no_grad():
        samp, nfe = SR.pc_sample(SR.OUVE(1.5, 0.05, 0.5, N=N),
                                 lambda x, t, y: -NR.ncsnpp_forward(sd, cfg, torch.cat([x, y], 1), t),
                                 Y, lambda: next(it), corrector_steps=steps, snr=0.5)
    xh_or = FR.spec_to_wav(samp, nfac, T0)
    check(f"{tag} enhance wav ({2 * N} evaluations, {backbone})", xh_or, xh_ref, 1e-4)
    fs = final["spec"].reshape(samp.shape) if "spec" in final else None
    if fs is not None:
        check(f"{tag} final sampler state", samp, fs, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), wav_in=ywav.numpy(), out=xh_ref.numpy(),
                        final_spec=c2np(fs if fs is not None else samp), nfe=np.array(N * (steps + 1)), N=np.array(N),
                        seeds=np.array([seed_w, seed_n, seed_wav]), noise_hash=np.array(nhash), sdhash=np.array(sd_hash(sd)),
                        noise_shape=np.array(list(Ysh)))


def gen_f13(ref):
    """F13: the product of the path at FULL width and FULL length of the sampler on a 1-s utterance (16 000 samples -> 126 -> 128
    frames).  This is what pins a bf16 / fp16 60-evaluation run against the REFERENCE instead of against the engine's own fp32 run."""
    _full_sampler_fixture(ref, "F13", "f13_full_sampler.npz", 16000, (11, 1313, 1301), "1-s utterance (about ten minutes)")


def gen_f14(ref):
    """F14: the same at the BENCH length - BASELINE.json configs[1]'s utterance: 4 s = 64 000 samples -> 501 -> 512 frames, i.e. the
    reference's own 60-evaluation `enhance` at the shape bench.py times (about 40 minutes on 8 cores: 2 x 60 CPU evaluations)."""
    _full_sampler_fixture(ref, "F14", "f14_full_sampler_4s.npz", 64000, (12, 1414, 1402), "4-s utterance = the bench length (about 40 minutes)")


def gen_f15(ref):
    """F15: BASELINE.json configs[3]'s SAMPLER on its network - `ncsnpplarge` (65.6 M, ncsnpp.py:460-470), N = 50 reverse steps + 1 `ald` corrector step
    each = 100 score evaluations (the configuration's "50-step PC + 1 corrector"), on a 2-s utterance (32 000 samples -> 251 -> 256 frames: the 8-s
    length of configs[3] would cost the CPU reference four hours; the forward at that length is pinned by F11).  About 25 minutes on 8 cores."""
    _full_sampler_fixture(ref, "F15", "f15_large_sampler.npz", 32000, (13, 1515, 1503), "ncsnpplarge, 2-s utterance, 100 evaluations (about 25 minutes)",
                          backbone="ncsnpplarge", N=50)


F16_LENGTHS = (156000, 158000, 160000)                    # 1219 / 1235 / 1251 frames: all in the 1280-frame bucket (util/other.py:102-109)
F16_SEEDS = dict(weights=16, x=1616, wav=1600, noise=1610, state=1620, ode_z=1630)
F16_ODE_TOL = 0.03


def gen_f16(ref):
    """F16: BASELINE.json configs[4] at its REAL shape - the seeded 27.8 M `ncsnpp` on 10-s rows (160 000 samples -> 1251 -> 1280
    frames; 256 x 1280 conv levels, L = 5120 attention: layerspp.py:82-86).  (a) NCSNpp.forward of ONE [1,2,256,1280] input
    (ncsnpp.py:281-450); (b) ScoreModel.enhance (model.py:273-310) with N = 3 reverse steps + 1 ald step each = 6 evaluations, wav -> wav,
    on THREE utterances of 156 000 / 158 000 / 160 000 samples, one by one as the reference runs them (they share the 1280-frame bucket),
    under recorded noise (regenerated from seeds on both sides); (c) one probability-flow right-hand side rsde.sde(x, t, y)[0]
    (sampling/__init__.py:104-106, sdes.py:123-145) at that shape; (d) ScoreModel.enhance(sampler_type="ode") of the 160 000-sample
    utterance with rtol = atol = F16_ODE_TOL (the solver's own loop at the real shape; the configured 1e-5 would cost the CPU reference
    ~500 evaluations of 20 s).  About an hour on 8 cores."""
    print("F16 configs[4] at its real shape: 27.8 M ncsnpp on 256 x 1280 (about an hour)")
    torch.set_num_threads(8)
    import time
    M, DM = ref["model"], ref["data_module"].SpecsDataModule
    S = F16_SEEDS
    cfg = NR.NCSNppConfig(**NR.NAMED_CONFIGS["ncsnpp"], input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=S["weights"])
    m = M.ScoreModel(backbone="ncsnpp", sde="ouve", data_module_cls=DM, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                     spec_factor=0.15, spec_abs_exponent=0.5)
    m.dnn.load_state_dict(sd)
    m.eval(no_ema=True)
    f16 = dict(sdhash=np.array(sd_hash(sd)), lengths=np.array(F16_LENGTHS), t=np.array([0.37], dtype=np.float32),
               seeds=np.array([S[k] for k in ("weights", "x", "wav", "noise", "state", "ode_z")]), ode_tol=np.array(F16_ODE_TOL))
    # (a) one forward
    xin = seeded_input((1, 2, 256, 1280), S["x"], 0.5)
    tt = torch.tensor([0.37])
    t0 = time.time()
    with torch.no_grad():
        y_ref = m.dnn(xin, tt)
        y_or = NR.ncsnpp_forward(sd, cfg, xin, tt)
    check("ncsnpp4 forward @ 256x1280", y_or, y_ref, 5e-5)
    print(f"  (a) two forwards {time.time() - t0:.0f} s")
    f16.update(fwd_y=c2np(y_ref), fwd_xhash=np.array(tensor_hash(xin)))
    # (b) three 6-evaluation enhance runs, one utterance per call
    N, steps = 3, 1
    orig = torch.randn_like
    nhashes = []
    for i, n in enumerate(F16_LENGTHS):
        t0 = time.time()
        ywav = torch.randn(1, n, generator=torch.Generator().manual_seed(S["wav"] + i)) * 0.1
        gn = torch.Generator().manual_seed(S["noise"] + i)
        noises = [SR.complex_randn((1, 1, 256, 1280), gn) for _ in range(1 + N * (steps + 1))]
        nhashes.append(hashlib.sha256(b"".join(c2np(z).tobytes() for z in noises)).hexdigest())
        it = iter(noises)
        final = {}
        to_audio_orig = m.to_audio

        def to_audio_spy(spec, length=None):
            final["spec"] = spec.detach().clone()
            return to_audio_orig(spec, length)
        torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
        m.to_audio = to_audio_spy
        try:
            with torch.no_grad():
                xh_ref = m.enhance(ywav.clone(), N=N, corrector="ald", corrector_steps=steps, snr=0.5)
        finally:
            torch.randn_like = orig
            m.to_audio = to_audio_orig
        f16[f"pc_out{i}"] = xh_ref.numpy()
        f16[f"pc_wavhash{i}"] = np.array(tensor_hash(ywav))
        if i == 0:                                         # the shortest row: the most padded frames behind its content
            f16["pc_final_spec0"] = c2np(final["spec"].reshape(1, 1, 256, 1280))
            Y, nfac, T0 = FR.wav_to_spec(ywav)
            it = iter(noises)
            with torch.no_grad():
                samp, nfe = SR.pc_sample(SR.OUVE(1.5, 0.05, 0.5, N=N), lambda x, t, y: -NR.ncsnpp_forward(sd, cfg, torch.cat([x, y], 1), t),
                                         Y, lambda: next(it), corrector_steps=steps, snr=0.5)
            check("F16 enhance wav (6 evaluations, 156 000 samples)", FR.spec_to_wav(samp, nfac, T0), xh_ref, 1e-4)
            check("F16 final sampler state", samp, final["spec"].reshape(samp.shape), 1e-4)
        print(f"  (b) utterance {i} ({n} samples) {time.time() - t0:.0f} s")
    f16["pc_noise_hashes"] = np.array(nhashes)
    f16.update(pc_N=np.array(N), pc_nfe=np.array(N * (steps + 1)))
    # (c) one probability-flow right-hand side at that shape
    ywav = torch.randn(1, F16_LENGTHS[2], generator=torch.Generator().manual_seed(S["wav"] + 2)) * 0.1
    Y, nfac, T0 = FR.wav_to_spec(ywav)
    xs = Y + seeded_input((1, 1, 256, 1280), S["state"], 0.2)
    tv = torch.tensor([0.5])
    sde = m.sde.copy()
    sde.N = 30
    with torch.no_grad():
        drift = sde.reverse(m, probability_flow=True).sde(xs, tv, Y)[0]
        s_or = -NR.ncsnpp_forward(sd, cfg, torch.cat([xs, Y], 1), tv)
        osde = SR.OUVE(1.5, 0.05, 0.5, N=30)
        f_or, g_or = osde.sde(xs, tv, Y)
        drift_or = f_or - 0.5 * g_or[:, None, None, None] ** 2 * s_or
    check("F16 probability-flow drift @ 256x1280", drift_or, drift, 5e-5)
    f16.update(pf_drift=c2np(drift), pf_t=np.array([0.5], dtype=np.float32), pf_xhash=np.array(tensor_hash(xs)))
    # (d) the ODE sampler's own loop at the real shape, loose tolerance
    t0 = time.time()
    zi = SR.complex_randn((1, 1, 256, 1280), torch.Generator().manual_seed(S["ode_z"]))
    torch.randn_like = lambda x, *a, **k: zi.to(x.dtype)
    try:
        with torch.no_grad():
            xh, nfe, _ = m.enhance(ywav.clone(), sampler_type="ode", timeit=True, device="cpu", rtol=F16_ODE_TOL, atol=F16_ODE_TOL)
    finally:
        torch.randn_like = orig
    nfe = int(nfe[0]) if isinstance(nfe, (list, tuple)) else int(nfe)
    print(f"  (d) enhance(ode, tol {F16_ODE_TOL}) nfev {nfe}, {time.time() - t0:.0f} s")
    f16.update(ode_out=xh.numpy(), ode_nfe=np.array(nfe), ode_zhash=np.array(tensor_hash(zi)))
    np.savez_compressed(os.path.join(OUT, "f16_cfg4_shape.npz"), **f16)


UPFIRDN_CASES = {
    # name: (N, H, W, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    "up2": (3, 6, 9, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1),           # upsample_2d(k=[1,3,3,1], factor 2): pads (2, 1) (up_or_down_sampling.py:219-224)
    "down2": (3, 6, 10, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1),        # downsample_2d: pads (1, 1) (up_or_down_sampling.py:252-257)
    "mixed": (2, 5, 7, 3, 5, 3, 1, 2, 1, -1, 2, 0, 3),        # factors / pads that differ per axis, a cropping (negative) pad, a 3 x 5 kernel
    "updown": (2, 4, 4, 2, 3, 2, 3, 3, 2, 1, 0, 2, 2),
}


def gen_f17(ref):
    """F17: the reference's one native-op seam, upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    (op/upfirdn2d.cpp:12-22), through its CPU form upfirdn2d_native (op/upfirdn2d.py:159-200) on four parameter sets: the two the
    network uses and two that no layer uses (storm_upfirdn2d accepts the full argument list)."""
    print("F17 upfirdn2d_native, four parameter sets")
    import importlib
    U = importlib.import_module("sgmse.backbones.ncsnpp_utils.op.upfirdn2d")
    g = torch.Generator().manual_seed(1717)
    f17 = {}
    for name, (N, H, W, kh, kw, ux, uy, dx, dy, px0, px1, py0, py1) in UPFIRDN_CASES.items():
        x = torch.randn(N, 1, H, W, generator=g)
        k = torch.randn(kh, kw, generator=g)
        if name in ("up2", "down2"):
            k1 = torch.tensor([1., 3., 3., 1.])
            k = torch.outer(k1, k1)
            k = k / k.sum() * (4 if name == "up2" else 1)
        y = U.upfirdn2d_native(x, k, ux, uy, dx, dy, px0, px1, py0, py1)[:, 0]
        check(f"upfirdn2d {name}", NR.upfirdn2d(x[:, 0], k, ux, uy, dx, dy, px0, px1, py0, py1), y, 1e-6)
        f17.update({f"{name}_x": x[:, 0].numpy(), f"{name}_k": k.numpy(), f"{name}_y": y.numpy(),
                    f"{name}_args": np.array([ux, uy, dx, dy, px0, px1, py0, py1])})
    np.savez_compressed(os.path.join(OUT, "f17_upfirdn2d.npz"), **f17)


def gen_f18(ref):
    """F18: the reference's second registered SDE, OUVPSDE (sdes.py:255-326).  (a) scalars: _beta / _std / _mean / sde / discretize /
    prior_sampling at three times for two parameter sets; (b) pc_sampler traces with recorded noise and an analytic score:
    reverse_diffusion + langevin, euler_maruyama + none, reverse_diffusion + none (the `ald` corrector rejects this SDE upstream,
    correctors.py:69 - asserted); (c) the ODE sampler (sampling/__init__.py:71-141) on one utterance; (d) ScoreModel(sde="ouvp").enhance
    wav -> wav with a tiny NCSN++ (reverse_diffusion + langevin, N = 4).  The oracle restatement (oracle/sde_ref.py::OUVP) is checked
    against every one of them on the way."""
    print("F18 OUVPSDE")
    f18 = {}
    g = torch.Generator().manual_seed(1818)
    orig = torch.randn_like
    for tag, (b0, b1, st) in OUVP_CASES.items():
        rs = ref["sdes"].OUVPSDE(beta_min=b0, beta_max=b1, stiffness=st, N=30)
        osde = SR.OUVP(b0, b1, st, N=30)
        tt = torch.tensor([1.0, 0.5, 0.03])
        xx = torch.randn(3, 1, 4, 4, dtype=torch.complex64, generator=g)
        yy = torch.randn(3, 1, 4, 4, dtype=torch.complex64, generator=g)
        zz = SR.complex_randn(xx.shape, g)
        std_ref, mean_ref = rs._std(tt), rs._mean(xx, tt, yy)
        drift_ref, diff_ref = rs.sde(xx, tt, yy)
        f_ref, G_ref = rs.discretize(xx, tt, yy)
        torch.randn_like = lambda x, *a, **k: zz.to(x.dtype)
        try:
            prior_ref = rs.prior_sampling(yy.shape, yy)
        finally:
            torch.randn_like = orig
        assert torch.equal(std_ref, osde.std(tt)) and torch.equal(mean_ref, osde.mean(xx, tt, yy))
        d_or, g_or = osde.sde(xx, tt, yy)
        assert torch.equal(drift_ref, d_or) and torch.equal(diff_ref, g_or)
        f_or, G_or = osde.discretize(xx, tt, yy)
        assert torch.equal(f_ref, f_or) and torch.equal(G_ref, G_or)
        assert torch.equal(prior_ref, osde.prior(yy, zz))
        f18.update({f"{tag}_t": tt.numpy(), f"{tag}_x": c2np(xx), f"{tag}_y": c2np(yy), f"{tag}_z": c2np(zz), f"{tag}_std": std_ref.numpy(),
                    f"{tag}_mean": c2np(mean_ref), f"{tag}_drift": c2np(drift_ref), f"{tag}_diff": diff_ref.numpy(), f"{tag}_f": c2np(f_ref),
                    f"{tag}_G": G_ref.numpy(), f"{tag}_prior": c2np(prior_ref)})

        def analytic_score(x, t, y, rs=rs):
            return -(x - y) / (rs._std(t)[:, None, None, None] ** 2 + 0.1)

        ysam = torch.randn(2, 1, 8, 16, dtype=torch.complex64, generator=g) * 0.3
        f18[f"{tag}_sam_y"] = c2np(ysam)
        for stag, (N, pred, corr, steps) in OUVP_SAMPLERS.items():
            ndraw = 1 + N * ((0 if corr == "none" else steps) + 1)
            noises = [SR.complex_randn(ysam.shape, g) for _ in range(ndraw)]
            it = iter(noises)
            torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
            try:
                sde = ref["sdes"].OUVPSDE(beta_min=b0, beta_max=b1, stiffness=st, N=N)
                x_ref, nfe_ref = ref["sampling"].get_pc_sampler(pred, corr, sde=sde, score_fn=analytic_score, y=ysam, eps=0.03, snr=0.5,
                                                                corrector_steps=steps)()
            finally:
                torch.randn_like = orig
            it = iter(noises)
            x_or, nfe_or = SR.pc_sample(SR.OUVP(b0, b1, st, N=N), analytic_score, ysam, lambda: next(it), predictor=pred, corrector=corr,
                                        corrector_steps=steps, snr=0.5)
            assert nfe_ref == nfe_or
            print(f"  {tag} sampler {stag}: nfe {nfe_ref}, bit-exact vs reference: {torch.equal(x_ref, x_or)}, rel {rel_l2(x_or, x_ref):.2e}")
            assert rel_l2(x_or, x_ref) < 1e-6
            f18.update({f"{tag}_{stag}_noise": np.stack([c2np(n) for n in noises]), f"{tag}_{stag}_out": c2np(x_ref),
                        f"{tag}_{stag}_nfe": np.array(nfe_ref)})
        try:                                                # upstream's ald corrector takes OUVE only
            ref["sampling"].get_pc_sampler("reverse_diffusion", "ald", sde=rs, score_fn=analytic_score, y=ysam)
            raise AssertionError("ald accepted an OUVPSDE")
        except NotImplementedError:
            pass

        # probability-flow right-hand side and the ODE sampler, one utterance (scipy RK45, rtol = atol = 1e-5)
        rsde = rs.reverse(analytic_score, probability_flow=True)
        pf_ref = rsde.sde(xx, tt, yy)[0]
        assert torch.equal(pf_ref, SR.pf_drift(osde, analytic_score, xx, tt, yy))
        f18[f"{tag}_pf"] = c2np(pf_ref)
        yo = ysam[:1]
        zo = SR.complex_randn(yo.shape, g)
        torch.randn_like = lambda x, *a, **k: zo.to(x.dtype)
        try:
            x_ode, nfe_ode = ref["sampling"].get_ode_sampler(rs, analytic_score, y=yo, eps=0.03, device="cpu")()
        finally:
            torch.randn_like = orig
        print(f"  {tag} ode: nfev {nfe_ode}")
        f18.update({f"{tag}_ode_z": c2np(zo), f"{tag}_ode_out": c2np(x_ode), f"{tag}_ode_nfe": np.array(nfe_ode)})

    # ScoreModel(sde="ouvp").enhance wav -> wav, tiny NCSN++
    M, DM = ref["model"], ref["data_module"].SpecsDataModule
    b0, b1, st = OUVP_CASES["a"]
    m = M.ScoreModel(backbone="ncsnpp", sde="ouvp", data_module_cls=DM, beta_min=b0, beta_max=b1, stiffness=st, spec_factor=0.15,
                     spec_abs_exponent=0.5, nf=8)
    cfg = NR.NCSNppConfig(nf=8, input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=81)
    m.dnn.load_state_dict(sd)
    m.eval(no_ema=True)
    wav = torch.randn(1, 8000, generator=g) * 0.1
    N = 4
    noises = [SR.complex_randn((1, 1, 256, 64), g) for _ in range(1 + 2 * N)]
    it = iter(noises)
    torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
    try:
        with torch.no_grad():
            xh, nfe, _ = m.enhance(wav.clone(), predictor="reverse_diffusion", corrector="langevin", N=N, corrector_steps=1, snr=0.5,
                                   timeit=True, device="cpu")
    finally:
        torch.randn_like = orig
    Y, nf_, _ = FR.wav_to_spec(wav, 0.15, 0.5)
    it = iter(noises)
    with torch.no_grad():
        samp, nfe_or = SR.pc_sample(SR.OUVP(b0, b1, st, N=N), lambda x, t, y: -NR.ncsnpp_forward(sd, cfg, torch.cat([x, y], 1), t), Y,
                                    lambda: next(it), predictor="reverse_diffusion", corrector="langevin", corrector_steps=1, snr=0.5)
    w_or = FR.istft(FR.spec_back(samp.squeeze(), 0.15, 0.5), wav.shape[1]) * nf_
    check("ouvp enhance tiny net", w_or, xh, 1e-4)
    f18.update(enh_wav=wav.numpy(), enh_noise=np.stack([c2np(n) for n in noises]), enh_out=xh.numpy(), enh_nfe=np.array(int(nfe)),
               enh_sdhash=np.array(sd_hash(sd)))
    np.savez_compressed(os.path.join(OUT, "f18_ouvp.npz"), **f18)


def gen_f19(ref):
    """F19: the two small registered variants of the backbone, `ncsnpp12M` and `ncsnpp6M` (ncsnpp.py:479-513: nf = 96, one ResNet block per
    level, no attention) - the reference's OWN classes (their hyper-parameters are part of what is pinned), seeded weights, one forward at
    32 x 64 (simulator-sized) and one at the real 256 x 128 (a 1-s utterance) with a batch of two."""
    print("F19 ncsnpp12M / ncsnpp6M")
    f19 = {}
    for name in ("ncsnpp12M", "ncsnpp6M"):
        cfg = NR.NCSNppConfig(**NR.NAMED_CONFIGS[name], input_channels=4)
        sd = NR.seeded_state_dict(cfg, seed=19)
        net = getattr(ref["ncsnpp"], "NCSNpp12M" if name == "ncsnpp12M" else "NCSNpp6M")(input_channels=4)
        net.load_state_dict(sd, strict=True)
        net.eval()
        nparam = sum(p.numel() for p in net.parameters())
        print(f"  {name}: {nparam / 1e6:.3f} M params")
        f19.update({f"{name}_sdhash": np.array(sd_hash(sd)), f"{name}_nparam": np.array(nparam)})
        for tag, (B, F, T) in F19_SHAPES.items():
            x, t = f19_inputs(name, tag)
            with torch.no_grad():
                y_ref = net(x, t)
                y_or = NR.ncsnpp_forward(sd, cfg, x, t)
            check(f"{name} forward {tag}", y_or, y_ref, 5e-5)
            f19.update({f"{name}_{tag}_xhash": np.array(hashlib.sha256(c2np(x).tobytes()).hexdigest()), f"{name}_{tag}_t": t.numpy(),
                        f"{name}_{tag}_y": c2np(y_ref)})
    np.savez_compressed(os.path.join(OUT, "f19_small_nets.npz"), **f19)


def gen_f20(ref):
    """F20: the data module's other settings (data_module.py:19-25, 142-148, 182-223; CLI --window / --n_fft / --hop_length / --spec_factor /
    --spec_abs_exponent): sqrt-Hann window, no magnitude compression (exponent 1), another exponent / factor, a 254-point and a
    256-hop transform - spec_fwd(stft(y)) and istft(spec_back(.)) of the reference's SpecsDataModule on one 3000-sample signal."""
    print("F20 data-module settings")
    DM = ref["data_module"].SpecsDataModule
    y = torch.randn(1, 3000, generator=torch.Generator().manual_seed(2020)) * 0.1
    f20 = dict(y=y.numpy())
    for tag, kw in F20_CASES.items():
        dm = DM(gpu=False, **kw)
        Y = dm.spec_fwd(dm.stft(y))
        w = dm.istft(dm.spec_back(Y), 3000)
        k = dict(n_fft=kw.get("n_fft", 510), hop=kw.get("hop_length", 128), kind=kw.get("window", "hann"))
        fac, e = kw.get("spec_factor", 0.15), kw.get("spec_abs_exponent", 0.5)
        Yo = FR.spec_fwd(FR.stft(y, **k), fac, e)
        assert torch.equal(Yo, Y) and torch.equal(FR.istft(FR.spec_back(Yo, fac, e), 3000, **k), w), tag
        f20.update({f"{tag}_Y": c2np(Y), f"{tag}_wav": w.numpy()})
    np.savez_compressed(os.path.join(OUT, "f20_data_module.npz"), **f20)


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-f9" in sys.argv:
        gen_f9(import_reference())
        return
    if "--only-f7" in sys.argv:
        gen_f7(import_reference())
        return
    if "--only-f8" in sys.argv:
        gen_f8(import_reference())
        return
    for flag, fn in (("--only-f10", gen_f10), ("--only-f11", gen_f11), ("--only-f12", gen_f12), ("--only-f13", gen_f13), ("--only-f14", gen_f14), ("--only-f15", gen_f15), ("--only-f16", gen_f16), ("--only-f17", gen_f17), ("--only-f18", gen_f18), ("--only-f19", gen_f19), ("--only-f20", gen_f20)):
        if flag in sys.argv:
            fn(import_reference())
            return
    torch.set_num_threads(8)
    torch.manual_seed(0)
    ref = import_reference()
    L, UD = ref["layerspp"], ref["updown"]
    g = torch.Generator().manual_seed(1234)

    # ---------------- F1: per-op ----------------
    print("F1 per-op")
    f1 = {}
    x = torch.randn(2, 5, 8, 12, generator=g)
    up_ref, dn_ref = UD.upsample_2d(x, [1, 3, 3, 1], factor=2), UD.downsample_2d(x, [1, 3, 3, 1], factor=2)
    check("fir_up2", NR.fir_up2(x), up_ref, 1e-6)
    check("fir_down2", NR.fir_down2(x), dn_ref, 1e-6)
    f1.update(fir_x=x.numpy(), fir_up=up_ref.numpy(), fir_down=dn_ref.numpy())
    for C in (8, 128, 384):
        xx = torch.randn(2, C, 8, 16, generator=g) * 1.5 + 0.3
        w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        gn = torch.nn.GroupNorm(min(C // 4, 32), C, eps=1e-6)
        gn.weight.data, gn.bias.data = w, b
        y_ref = torch.nn.SiLU()(gn(xx)).detach()
        check(f"gn_silu C={C}", NR.silu(NR.group_norm(xx, w, b)), y_ref, 1e-6)
        f1.update({f"gn{C}_x": xx.numpy(), f"gn{C}_w": w.numpy(), f"gn{C}_b": b.numpy(), f"gn{C}_y": y_ref.numpy()})
    # attention block
    C = 16
    blk = L.AttnBlockpp(channels=C, skip_rescale=True, init_scale=0.)
    sd = {k: (torch.randn(v.shape, generator=g) * (0.3 if "NIN" in k and k.endswith("W") else 0.1)
              + (1.0 if "GroupNorm_0.weight" in k else 0.0)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    xa = torch.randn(2, C, 4, 8, generator=g)
    ya = blk(xa).detach()
    check("attnblock", NR.attnblock(NR._SD(sd), xa), ya, 1e-5)
    f1.update(attn_x=xa.numpy(), attn_y=ya.numpy(), **{"attn_" + k: v.numpy() for k, v in sd.items()})
    # resnet blocks plain / up / down
    for tag, kw in (("plain", {}), ("up", dict(up=True)), ("down", dict(down=True)), ("widen", dict(out_ch=24))):
        blk = L.ResnetBlockBigGANpp(act=torch.nn.SiLU(), in_ch=16, temb_dim=32, dropout=0., fir=True,
                                    fir_kernel=[1, 3, 3, 1], init_scale=0., skip_rescale=True, **kw)
        sd = {k: torch.randn(v.shape, generator=g) * 0.15 + (1.0 if "GroupNorm" in k and k.endswith("weight") else 0.0)
              for k, v in blk.state_dict().items()}
        blk.load_state_dict(sd)
        blk.eval()
        xr, te = torch.randn(2, 16, 8, 16, generator=g), torch.randn(2, 32, generator=g)
        yr = blk(xr, te).detach()
        check(f"resblock {tag}", NR.resblock(NR._SD(sd), xr, te, up=kw.get("up", False), down=kw.get("down", False)), yr, 1e-5)
        f1.update({f"res_{tag}_x": xr.numpy(), f"res_{tag}_temb": te.numpy(), f"res_{tag}_y": yr.numpy(),
                   **{f"res_{tag}_" + k: v.numpy() for k, v in sd.items()}})
    np.savez_compressed(os.path.join(OUT, "f1_ops.npz"), **f1)

    # ---------------- F2: tiny nets ----------------
    print("F2 tiny nets")
    f2 = {}
    t = torch.tensor([0.9, 0.05])
    for name, (ckw, (F_, T_)) in TINY.items():
        cfg = NR.NCSNppConfig(**ckw)
        sd = NR.seeded_state_dict(cfg, seed=7)
        net = ref_backbone(ref, cfg, sd)
        xin = torch.randn(2, cfg.in_ch // 2, F_, T_, dtype=torch.complex64, generator=g) * 0.5
        with torch.no_grad():
            y_ref = net(xin, None if cfg.discriminative else t)
            y_or = NR.ncsnpp_forward(sd, cfg, xin, t)
        check(f"{name} forward", y_or, y_ref, 2e-5)
        f2.update({f"{name}_x": c2np(xin), f"{name}_y": c2np(y_ref), f"{name}_sdhash": np.array(sd_hash(sd))})
    f2["t"] = t.numpy()
    np.savez_compressed(os.path.join(OUT, "f2_tiny_nets.npz"), **f2)

    # ---------------- F3: full-width net ----------------
    print("F3 full-width nets (this takes a minute)")
    f3 = {}
    for name, ckw, shape in (("ncsnpp4", dict(input_channels=4), (256, 64)),
                             ("ncsnpp6", dict(input_channels=6), (64, 64)),
                             ("ncsnpp2d", dict(input_channels=2, discriminative=True), (64, 64)),
                             ("large4", dict(**NR.NAMED_CONFIGS["ncsnpplarge"], input_channels=4), (256, 64))):
        cfg = NR.NCSNppConfig(**ckw)
        sd = NR.seeded_state_dict(cfg, seed=11)
        net = ref_backbone(ref, cfg, sd)
        nparam = sum(v.numel() for k, v in sd.items())
        xin = torch.randn(1, cfg.in_ch // 2, *shape, dtype=torch.complex64, generator=g) * 0.5
        tt = torch.tensor([0.37])
        with torch.no_grad():
            y_ref = net(xin, None if cfg.discriminative else tt)
            y_or = NR.ncsnpp_forward(sd, cfg, xin, tt)
        print(f"  {name}: {nparam/1e6:.3f} M params")
        check(f"{name} forward", y_or, y_ref, 5e-5)
        f3.update({f"{name}_x": c2np(xin), f"{name}_y": c2np(y_ref), f"{name}_sdhash": np.array(sd_hash(sd)),
                   f"{name}_nparam": np.array(nparam)})
    f3["t"] = np.array([0.37], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "f3_full_nets.npz"), **f3)

    # ---------------- F4: SDE scalars + sampler traces ----------------
    print("F4 SDE / sampler")
    f4 = {}
    rs = ref["sdes"].OUVESDE(theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30)
    osde = SR.OUVE(1.5, 0.05, 0.5, N=30)
    tt = torch.tensor([1.0, 0.5, 0.03])
    std_ref = rs._std(tt)
    assert torch.equal(std_ref, osde.std(tt))
    xx = torch.randn(3, 1, 4, 4, dtype=torch.complex64, generator=g)
    yy = torch.randn(3, 1, 4, 4, dtype=torch.complex64, generator=g)
    f_ref, G_ref = rs.discretize(xx, tt, yy)
    f_or, G_or = osde.discretize(xx, tt, yy)
    assert torch.equal(f_ref, f_or) and torch.equal(G_ref, G_or)
    f4.update(t=tt.numpy(), std=std_ref.numpy(), disc_x=c2np(xx), disc_y=c2np(yy), disc_f=c2np(f_ref), disc_G=G_ref.numpy())

    def analytic_score(x, t, y):
        return -(x - y) / (rs._std(t)[:, None, None, None] ** 2 + 0.1)

    def run_ref_sampler(score_fn, y, N, predictor, corrector, steps, snr, noises, conditioning=None):
        it = iter(noises)
        orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
        try:
            sde = ref["sdes"].OUVESDE(theta=1.5, sigma_min=0.05, sigma_max=0.5, N=N)
            sampler = ref["sampling"].get_pc_sampler(predictor, corrector, sde=sde, score_fn=score_fn, y=y,
                                                     eps=0.03, snr=snr, corrector_steps=steps,
                                                     conditioning=conditioning)
            return sampler()
        finally:
            torch.randn_like = orig

    ysam = torch.randn(2, 1, 8, 16, dtype=torch.complex64, generator=g) * 0.3
    for tag, (N, pred, corr, steps) in dict(ald2=(7, "reverse_diffusion", "ald", 2),
                                             lang=(5, "reverse_diffusion", "langevin", 1),
                                             em=(6, "euler_maruyama", "none", 1),
                                             none=(4, "reverse_diffusion", "none", 1)).items():
        ndraw = 1 + N * ((0 if corr == "none" else steps) + 1)
        noises = [SR.complex_randn(ysam.shape, g) for _ in range(ndraw)]
        x_ref, nfe_ref = run_ref_sampler(analytic_score, ysam, N, pred, corr, steps, 0.5, noises)
        it = iter(noises)
        x_or, nfe_or = SR.pc_sample(SR.OUVE(1.5, 0.05, 0.5, N=N), analytic_score, ysam, lambda: next(it),
                                    predictor=pred, corrector=corr, corrector_steps=steps, snr=0.5)
        assert nfe_ref == nfe_or, (nfe_ref, nfe_or)
        bit = torch.equal(x_ref, x_or)
        print(f"  sampler {tag}: nfe {nfe_ref}, bit-exact vs reference: {bit}, rel {rel_l2(x_or, x_ref):.2e}")
        assert rel_l2(x_or, x_ref) < 1e-6
        f4.update({f"{tag}_noise": np.stack([c2np(n) for n in noises]), f"{tag}_out": c2np(x_ref), f"{tag}_nfe": np.array(nfe_ref)})
    f4["sam_y"] = c2np(ysam)

    # sampler with a tiny net as score function (ScoreModel.forward = -dnn(cat[x,y], t), model.py:127-132)
    cfg = NR.NCSNppConfig(**TINY["tiny4"][0])
    sd = NR.seeded_state_dict(cfg, seed=7)
    net = ref_backbone(ref, cfg, sd)
    ynet = torch.randn(2, 1, 32, 64, dtype=torch.complex64, generator=g) * 0.3
    N, steps = 3, 1
    noises = [SR.complex_randn(ynet.shape, g) for _ in range(1 + N * (steps + 1))]
    with torch.no_grad():
        x_ref, nfe = run_ref_sampler(lambda x, t, y: -net(torch.cat([x, y], 1), t), ynet, N, "reverse_diffusion", "ald", steps, 0.5, noises)
        it = iter(noises)
        x_or, _ = SR.pc_sample(SR.OUVE(1.5, 0.05, 0.5, N=N), lambda x, t, y: -NR.ncsnpp_forward(sd, cfg, torch.cat([x, y], 1), t),
                               ynet, lambda: next(it), corrector_steps=steps, snr=0.5)
    check("sampler tiny4 net", x_or, x_ref, 1e-4)
    f4.update(net_y=c2np(ynet), net_noise=np.stack([c2np(n) for n in noises]), net_out=c2np(x_ref), net_nfe=np.array(nfe))
    np.savez_compressed(os.path.join(OUT, "f4_sampler.npz"), **f4)

    # ---------------- F5: front/back end ----------------
    print("F5 front/back end")
    f5 = {}
    for Lsig in (8000, 64000):
        for fac in (0.15, 0.33):
            dm = ref["data_module"].SpecsDataModule(spec_factor=fac, spec_abs_exponent=0.5, gpu=False)
            y = torch.randn(1, Lsig, generator=torch.Generator().manual_seed(1234 + Lsig)) * 0.1
            nf_ = y.abs().max().item()
            Y_ref = ref["other"].pad_spec(torch.unsqueeze(dm.spec_fwd(dm.stft(y / nf_)), 0))
            Y_or, nf_or, _ = FR.wav_to_spec(y, fac, 0.5)
            assert torch.equal(Y_ref, Y_or) and nf_ == nf_or
            w_ref = dm.istft(dm.spec_back(Y_ref.squeeze()), Lsig)
            w_or = FR.istft(FR.spec_back(Y_or.squeeze(), fac, 0.5), Lsig)
            assert torch.equal(w_ref, w_or)
            key = f"L{Lsig}_f{int(fac*100)}"
            if Lsig == 8000:
                f5.update({f"{key}_y": y.numpy(), f"{key}_Y": c2np(Y_ref), f"{key}_wav": w_ref.numpy()})
            else:   # keep the big one small: a frame slice + the tail of the waveform
                f5.update({f"{key}_Yslice": c2np(Y_ref[..., 245:262]), f"{key}_wavtail": w_ref[..., -512:].numpy(),
                           f"{key}_wavhead": w_ref[..., :512].numpy()})
            print(f"  {key}: frames {Y_ref.shape[-1]}, istft rel err vs y/nf {rel_l2(w_ref[..., :Lsig-200], (y/nf_)[..., :Lsig-200]):.2e}")
    np.savez_compressed(os.path.join(OUT, "f5_frontend.npz"), **f5)

    # ---------------- F6: end-to-end enhance ----------------
    print("F6 enhance() wav -> wav")
    f6 = {}
    M = ref["model"]
    DM = ref["data_module"].SpecsDataModule
    ywav = torch.randn(1, 8000, generator=torch.Generator().manual_seed(99)) * 0.1
    common = dict(sde="ouve", data_module_cls=DM, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                  spec_factor=0.15, spec_abs_exponent=0.5, nf=8)
    # score-only
    m = M.ScoreModel(backbone="ncsnpp", **common)
    cfg = NR.NCSNppConfig(nf=8, input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=21)
    m.dnn.load_state_dict(sd)
    m.eval(no_ema=True)
    N, steps = 3, 1
    Ysh = FR.wav_to_spec(ywav)[0].shape
    noises = [SR.complex_randn(Ysh, g) for _ in range(1 + N * (steps + 1))]
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
    try:
        with torch.no_grad():
            xh_ref = m.enhance(ywav.clone(), N=N, corrector="ald", corrector_steps=steps, snr=0.5)
    finally:
        torch.randn_like = orig
    Y, nfac, T0 = FR.wav_to_spec(ywav)
    it = iter(noises)
    with torch.no_grad():
        samp, _ = SR.pc_sample(SR.OUVE(1.5, 0.05, 0.5, N=N),
                               lambda x, t, y: -NR.ncsnpp_forward(sd, cfg, torch.cat([x, y], 1), t),
                               Y, lambda: next(it), corrector_steps=steps, snr=0.5)
    xh_or = FR.spec_to_wav(samp, nfac, T0)
    check("enhance score-only", xh_or, xh_ref, 1e-4)
    f6.update(wav_in=ywav.numpy(), so_noise=np.stack([c2np(n) for n in noises]), so_out=xh_ref.numpy())
    # StoRM, the three conditioning variants (model.py:743-750)
    for cond in ("both", "noisy", "post_denoiser"):
        m = M.StochasticRegenerationModel(backbone_denoiser="ncsnpp", backbone_score="ncsnpp", condition=cond, **dict(common))
        cfg_d = NR.NCSNppConfig(nf=8, input_channels=2, discriminative=True)
        cfg_s = NR.NCSNppConfig(nf=8, input_channels=6 if cond == "both" else 4)
        sd_d, sd_s = NR.seeded_state_dict(cfg_d, seed=31), NR.seeded_state_dict(cfg_s, seed=32)
        m.denoiser_net.load_state_dict(sd_d)
        m.score_net.load_state_dict(sd_s)
        m.eval(no_ema=True)
        N = 3
        noises = [SR.complex_randn(Ysh, g) for _ in range(1 + N)]
        it = iter(noises)
        torch.randn_like = lambda x, *a, **k: next(it).to(x.dtype)
        try:
            with torch.no_grad():
                xh_ref = m.enhance(ywav.clone(), N=N, corrector="none", snr=0.5)
        finally:
            torch.randn_like = orig
        with torch.no_grad():
            Yd = NR.ncsnpp_forward(sd_d, cfg_d, Y, None)
            condl = dict(both=[Y, Yd], noisy=[Y], post_denoiser=[Yd])[cond]
            it = iter(noises)
            samp, _ = SR.pc_sample(SR.OUVE(1.5, 0.05, 0.5, N=N),
                                   lambda x, t, y: -NR.ncsnpp_forward(sd_s, cfg_s, torch.cat([x] + condl, 1), t),
                                   Yd, lambda: next(it), corrector="none", snr=0.5)
        xh_or = FR.spec_to_wav(samp, nfac, T0)
        check(f"enhance storm/{cond}", xh_or, xh_ref, 1e-4)
        f6.update({f"storm_{cond}_noise": np.stack([c2np(n) for n in noises]), f"storm_{cond}_out": xh_ref.numpy()})
    np.savez_compressed(os.path.join(OUT, "f6_enhance.npz"), **f6)

    gen_f7(ref)
    gen_f8(ref)
    gen_f9(ref)
    gen_f10(ref)
    gen_f11(ref)
    gen_f12(ref)
    gen_f13(ref)
    gen_f14(ref)
    gen_f15(ref)
    gen_f17(ref)
    gen_f19(ref)
    gen_f20(ref)
    gen_f18(ref)                                 # (F16, the 27.8 M net at 256 x 1280, is generated on request: --only-f16)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))/1024:.0f} KiB")
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
