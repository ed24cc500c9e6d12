"""PC sampler parity against the reference's own traces (tests/golden/f4_sampler.npz: recorded noise,
analytic and NCSN++ score functions) — host simulation on CPU, real kernels with -m gpu."""
import pytest
import torch

from oracle import ncsnpp_ref as NR
from oracle.make_golden import TINY
from tests.backend import dev  # noqa: F401
from tests.util import rel_l2

T = torch.from_numpy


@pytest.mark.parametrize("tag,N,pred,corr,steps", [("ald2", 7, "reverse_diffusion", "ald", 2),
                                                    ("lang", 5, "reverse_diffusion", "langevin", 1),
                                                    ("em", 6, "euler_maruyama", "none", 1),
                                                    ("none", 4, "reverse_diffusion", "none", 1)])
def test_pc_sampler_analytic_score(dev, golden, tag, N, pred, corr, steps):
    from storm_amd.sampling import get_pc_sampler
    from storm_amd.sdes import OUVESDE
    g = golden["f4_sampler"]
    sde = OUVESDE(1.5, 0.05, 0.5, N=N)

    def score(x, t, y):
        return -(x - y) / (sde._std(t)[:, None, None, None] ** 2 + 0.1)
    it = iter(T(g[f"{tag}_noise"]))
    sampler = get_pc_sampler(pred, corr, sde=sde, score_fn=score, y=T(g["sam_y"]).to(dev), eps=0.03, snr=0.5,
                             corrector_steps=steps, noise_fn=lambda: next(it))
    x, nfe = sampler()
    assert nfe == int(g[f"{tag}_nfe"])
    assert rel_l2(x.cpu(), g[f"{tag}_out"]) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 5e-2)])
def test_pc_sampler_with_net(dev, golden, dtype, tol):
    from storm_amd.backbones.ncsnpp import NCSNpp
    from storm_amd.sampling import get_pc_sampler
    from storm_amd.sdes import OUVESDE
    g = golden["f4_sampler"]
    kw = TINY["tiny4"][0]
    net = NCSNpp(**kw)
    net.load_state_dict(NR.seeded_state_dict(NR.NCSNppConfig(**kw), seed=7))
    net = net.to(dev).set_compute_dtype(dtype)
    it = iter(T(g["net_noise"]))
    sampler = get_pc_sampler("reverse_diffusion", "ald", sde=OUVESDE(1.5, 0.05, 0.5, N=3),
                             score_fn=lambda x, t, y: -net(torch.cat([x, y], 1), t), y=T(g["net_y"]).to(dev),
                             eps=0.03, snr=0.5, corrector_steps=1, noise_fn=lambda: next(it))
    x, nfe = sampler()
    assert nfe == int(g["net_nfe"])
    assert rel_l2(x.cpu(), g["net_out"]) < tol


def test_registries_and_errors():
    from storm_amd.sampling import CorrectorRegistry, PredictorRegistry
    from storm_amd.sdes import SDERegistry
    assert set(PredictorRegistry.get_all_names()) == {"euler_maruyama", "reverse_diffusion", "none"}
    assert set(CorrectorRegistry.get_all_names()) == {"langevin", "ald", "none"}
    assert "ouve" in SDERegistry.get_all_names()
    with pytest.raises(ValueError):
        SDERegistry.get_by_name("ouvesde")          # the reference's default string is not a registered name either


def test_philox_sampler_is_seeded(dev):
    """production path: noise generated in-kernel; same seed -> same sample, different seed -> different"""
    from storm_amd.sampling import get_pc_sampler
    from storm_amd.sdes import OUVESDE
    sde = OUVESDE(1.5, 0.05, 0.5, N=3)
    y = (torch.randn(2, 1, 8, 16, dtype=torch.complex64, generator=torch.Generator().manual_seed(0)) * 0.3).to(dev)
    score = lambda x, t, yy: -(x - yy)
    run = lambda seed: get_pc_sampler("reverse_diffusion", "ald", sde=sde, score_fn=score, y=y, snr=0.5, seed=seed)()[0].cpu()
    a, b, c = run(1), run(1), run(2)
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()


def test_ode_sampler_vs_reference(dev, golden):
    """device Dormand-Prince with scipy's step controller vs the reference's scipy.solve_ivp run"""
    from storm_amd.sampling import get_ode_sampler
    from storm_amd.sdes import OUVESDE
    g = golden["f7_ode"]
    sde = OUVESDE(1.5, 0.05, 0.5, N=30)

    def score(x, t, y):
        return -(x - y) / (sde._std(t)[:, None, None, None] ** 2 + 0.1)
    z = T(g["z"]).to(dev)
    sampler = get_ode_sampler(sde, score, y=T(g["y"]).to(dev), eps=0.03, noise_fn=lambda: z)
    x, nfe = sampler()
    assert abs(nfe - int(g["nfe"])) <= 12, (nfe, int(g["nfe"]))
    assert rel_l2(x.cpu(), g["out"]) < 1e-3


@pytest.mark.gpu
def test_bf16_sampler_drift_over_a_full_run():
    """BASELINE.json configs[1] numerics: the FULL 30-step PC run (reverse_diffusion + 1 ald step = 60 score evaluations of
    the 27.8 M network, 4-s utterances) in the bench precision (bf16 MFMA operands and activations) against the fp32 engine
    (itself 1e-6 from the oracle) under the SAME noise (in-kernel Philox stream of the same seed): the error a score
    evaluation makes (1e-2) is amplified by 1/t in the network head and re-enters 59 times.  wav rel-L2 <= 5e-2."""
    import bench
    from tests.backend import setup_backend
    from storm_amd.model import ScoreModel
    dev = setup_backend("hip")
    model = ScoreModel(backbone="ncsnpp", sde="ouve", theta=1.5, sigma_min=0.05, sigma_max=0.5, spec_factor=0.15,
                       spec_abs_exponent=0.5)
    bench.randomize(model, seed=0)
    model._error_loading_ema = True
    model.eval()
    model = model.to(dev)
    wav = (0.1 * torch.randn(2, 64000, generator=torch.Generator().manual_seed(3))).to(dev)
    outs = {}
    for prec in ("fp32", "bf16"):
        model.set_precision(prec)
        x, nfe = model.enhance_batch(wav, predictor="reverse_diffusion", corrector="ald", N=30, corrector_steps=1, snr=0.5,
                                     seed=7, return_nfe=True)
        assert nfe == 60
        outs[prec] = x.float().cpu()
    err = rel_l2(outs["bf16"], outs["fp32"])
    print(f"60-NFE PC run, bf16 vs fp32 engine under identical noise: wav rel-L2 {err:.3e}")
    assert err < 5e-2
