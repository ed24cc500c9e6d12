"""Model wrappers — drop-in for the inference surface of sgmse/model.py (ScoreModel :24,
DiscriminativeModel :320, StochasticRegenerationModel :392) without Lightning: ``enhance()``,
``get_pc_sampler()``, ``get_ode_sampler()``, ``forward`` / ``forward_score`` / ``forward_denoiser``,
``to_audio`` / ``_stft`` / ``_istft`` / ``_forward_transform`` / ``_backward_transform``,
``eval(no_ema=False)`` EMA swap and ``load_from_checkpoint``.  Training methods are out of scope.

New surface (not in the reference): ``enhance_batch`` (several utterances per call, equal to
per-utterance ``enhance`` calls), ``set_precision`` and the ``noise_fn`` / ``seed`` sampler knobs.
"""
import time
import warnings
from math import ceil

import torch
import torch.nn as nn

from . import sampling
from .backbones import BackboneRegistry
from .checkpoint import load_checkpoint_file
from .data_module import SpecsDataModule
from .sdes import SDERegistry

_PRECISIONS = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
               "fp16": torch.float16, "float16": torch.float16, "half": torch.float16}


class _EMA:
    """Inference-side mirror of torch_ema.ExponentialMovingAverage (store / copy_to / restore /
    load_state_dict); the shadow parameters come from the checkpoint's ``ema`` entry (model.py:86-108)."""

    def __init__(self, parameters, decay):
        self.decay = decay
        self.shadow_params = None
        self.collected_params = None

    def load_state_dict(self, sd):
        self.decay = sd.get("decay", self.decay)
        self.shadow_params = [p.clone() for p in sd["shadow_params"]]

    def state_dict(self):
        return {"decay": self.decay, "shadow_params": self.shadow_params, "collected_params": self.collected_params}

    def _match(self, parameters):
        params = list(parameters)
        if self.shadow_params is None:
            return None, params
        if len(self.shadow_params) == len(params):
            return self.shadow_params, params
        trainable = [p for p in params if p.requires_grad]
        if len(self.shadow_params) == len(trainable):
            return self.shadow_params, trainable
        raise RuntimeError(f"EMA state has {len(self.shadow_params)} tensors, model has {len(params)} parameters")

    def store(self, parameters):
        self.collected_params = [p.detach().clone() for p in parameters]

    def copy_to(self, parameters):
        shadow, params = self._match(parameters)
        if shadow is None:
            return
        with torch.no_grad():
            for s, p in zip(shadow, params):
                p.copy_(s.to(p.device))

    def restore(self, parameters):
        if self.collected_params is None:
            return
        with torch.no_grad():
            for c, p in zip(self.collected_params, parameters):
                p.copy_(c)
        self.collected_params = None

    def to(self, *a, **k):
        pass


class _Base(nn.Module):
    """Shared inference plumbing of the three model classes."""

    def _init_common(self, sde, t_eps, ema_decay, data_module_cls, kwargs):
        sde_cls = SDERegistry.get_by_name(sde)
        self.sde = sde_cls(**kwargs)
        self.t_eps = t_eps
        self.ema_decay = ema_decay
        self._error_loading_ema = False
        dm_cls = data_module_cls if data_module_cls is not None else SpecsDataModule
        self.data_module = dm_cls(**kwargs, gpu=kwargs.get("gpus", 0) > 0)

    def _backbones(self):
        return [m for m in self.children() if hasattr(m, "set_compute_dtype")]

    def set_precision(self, precision):
        """'fp32': exact-fp32 MFMA path (reference numerics); 'bf16' / 'fp16': 16-bit MFMA operands / activations
        with fp32 accumulation, statistics, time embedding and SDE state."""
        dt = _PRECISIONS[precision] if isinstance(precision, str) else precision
        for m in self._backbones():
            m.set_compute_dtype(dt)
        return self

    # ---- EMA swap (model.py:97-111) ----------------------------------------------------------
    def on_load_checkpoint(self, checkpoint):
        ema = checkpoint.get("ema", None)
        if ema is not None:
            self.ema.load_state_dict(ema)
        else:
            self._error_loading_ema = True
            warnings.warn("EMA state_dict not found in checkpoint!")

    def train(self, mode=True, no_ema=False):
        res = super().train(mode)
        if not self._error_loading_ema:
            if mode is False and not no_ema:
                self.ema.store(self.parameters())
                self.ema.copy_to(self.parameters())
            else:
                if self.ema.collected_params is not None:
                    self.ema.restore(self.parameters())
            for m in self._backbones():
                m.invalidate()
        return res

    def eval(self, no_ema=False):
        return self.train(False, no_ema=no_ema)

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", **overrides):
        """Lightning-free reader of a pytorch-lightning 1.8 ``.ckpt`` (enhancement.py:56-59): builds the
        model from ``hyper_parameters`` (+ overrides), loads ``state_dict`` and the ``ema`` shadow weights."""
        ckpt = load_checkpoint_file(checkpoint_path, map_location)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        if not isinstance(hp.get("data_module_cls"), type):
            hp["data_module_cls"] = SpecsDataModule
        model = cls(**hp)
        missing, unexpected = model.load_state_dict(ckpt["state_dict"], strict=False)
        if missing:
            raise RuntimeError(f"checkpoint lacks parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        model.on_load_checkpoint(ckpt)
        return model

    # ---- audio <-> spectrogram (model.py:258-271) ---------------------------------------------
    def to_audio(self, spec, length=None):
        return self._istft(self._backward_transform(spec), length)

    def _forward_transform(self, spec):
        return self.data_module.spec_fwd(spec)

    def _backward_transform(self, spec):
        return self.data_module.spec_back(spec)

    def _stft(self, sig):
        return self.data_module.stft(sig)

    def _istft(self, spec, length=None):
        return self.data_module.istft(spec, length)

    def _prepare(self, y, lengths=None):
        """model.py:282-286 for a batch: y [B, L] host/device -> (Y [B,1,F,Tpad], peak [B], L).  lengths: the rows' own sample
        counts when utterances of different lengths share the batch (same padded frame count; rows zero filled)."""
        if y.dim() != 2:
            raise ValueError("expected a waveform batch [B, L]")
        yd = y.to(device=self.device, dtype=torch.float32).contiguous()
        Y, peak = self.data_module.wav_to_spec(yd, pad_to=64, lengths=lengths)
        return Y, peak, y.size(1)

    def _score_network(self):
        """the network whose evaluations a grouped stream shares (ScoreModel: dnn; StoRM: score_net - its denoiser runs once per micro-batch)"""
        return getattr(self, "score_net", None) or self.dnn

    def enhance_stream(self, batches, grouped=True, return_nfe=False, seed=None, seeds=None, noise_fns=None, width=None, **kwargs):
        """(ScoreModel and StochasticRegenerationModel.)  A stream of ragged micro-batches (BASELINE.json configs[4]) in lockstep: `batches` = [(y [b, L], lengths or None), ...] as
        storm_amd.distributed.bucket_by_frames forms them.  Every micro-batch runs enhance_batch - the same sampler, noise stream and
        (ODE) per-row step control as its own call - but the score evaluations of all micro-batches that are still running share ONE
        grouped network call per step (storm_amd.sampling.grouped, storm_ncsnpp_forward_group): the layers with a grouped kernel see
        the whole stream's pixel tiles in one launch instead of 2 - 3 rows at a time.  grouped=False: one micro-batch after the other.
        seed: micro-batch k draws from the Philox stream seed + k (seeds: one seed per micro-batch; noise_fns: one injected-noise callable per micro-batch instead).
        width: at most this many micro-batches in flight (None = all); a finished one is replaced by the next of the list (storm_amd.sampling.grouped.run_grouped).
        Returns the list of enhanced batches (and the mean evaluations per utterance with return_nfe); self.last_nfev_stream = the
        evaluations every micro-batch executed."""
        from .sampling.grouped import run_grouped
        outs = [None] * len(batches)

        def one(k):
            yb, bl = batches[k]
            kw = dict(kwargs)
            if seeds is not None:
                kw["seed"] = seeds[k]
            elif seed is not None:
                kw["seed"] = seed + k
            if noise_fns is not None:
                kw["noise_fn"] = noise_fns[k]                  # (parity runs: the draws of micro-batch k)
            return self.enhance_batch(yb, lengths=bl, return_nfe=True, **kw)
        fns = [(lambda k=k: one(k)) for k in range(len(batches))]
        res, batcher = run_grouped(self._score_network(), fns, device=self.device, width=width) if grouped else ([f() for f in fns], None)
        outs = [r[0] for r in res]
        self.last_nfev_stream = [r[1] for r in res]
        self.last_group_calls = None if batcher is None else (batcher.calls, batcher.rows)
        if return_nfe:
            rows = sum(b[0].shape[0] for b in batches)
            return outs, sum(r[1] * b[0].shape[0] for r, b in zip(res, batches)) / rows
        return outs

    def _sampler_minibatched(self, make, y, minibatch):
        M = y.shape[0]

        def batched_sampling_fn():
            samples, ns = [], []
            for i in range(int(ceil(M / minibatch))):
                sample, n = make(slice(i * minibatch, (i + 1) * minibatch))()
                samples.append(sample)
                ns.append(n)
            return torch.cat(samples, dim=0), ns
        return batched_sampling_fn


class ScoreModel(_Base):
    def __init__(self, backbone: str = "ncsnpp", sde: str = "ouve", lr: float = 1e-4, ema_decay: float = 0.999,
                 t_eps: float = 3e-2, transform: str = "none", nolog: bool = False, num_eval_files: int = 50,
                 loss_type: str = "mse", data_module_cls=None, **kwargs):
        super().__init__()
        dnn_cls = BackboneRegistry.get_by_name(backbone)
        kwargs.update(input_channels=4)                                    # model.py:47
        self.dnn = dnn_cls(**kwargs)
        self._init_common(sde, t_eps, ema_decay, data_module_cls, kwargs)
        self.ema = _EMA(self.parameters(), decay=ema_decay)
        self.lr, self.loss_type, self.num_eval_files, self.nolog = lr, loss_type, num_eval_files, nolog
        self._set_score_sign()

    def _set_score_sign(self):
        if hasattr(self.dnn, "negate_output"):
            self.dnn.negate_output = True          # score = -dnn(...) (model.py:131-132) folded into the output head

    def _raw_dnn_output(self, x, t, y):
        return -self.forward(x, t, y)

    def forward(self, x, t, y, **kwargs):
        """score = -dnn(cat[x, y], t); x, y complex64 [B,1,F,T], t [B]."""
        return self.dnn.forward_parts([x[:, 0], y[:, 0]], t)

    def get_pc_sampler(self, predictor_name, corrector_name, y, N=None, minibatch=None, scale_factor=None, **kwargs):
        N = self.sde.N if N is None else N
        sde = self.sde.copy()
        sde.N = N
        kwargs = {"eps": self.t_eps, **kwargs}
        if minibatch is None:
            return sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y, **kwargs)
        return self._sampler_minibatched(
            lambda sl: sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y[sl], **kwargs),
            y, minibatch)

    def get_ode_sampler(self, y, N=None, minibatch=1, **kwargs):
        N = self.sde.N if N is None else N
        sde = self.sde.copy()
        sde.N = N
        kwargs = {"eps": self.t_eps, **kwargs}
        if minibatch is None:
            return sampling.get_ode_sampler(sde, self, y=y, **kwargs)
        return self._sampler_minibatched(lambda sl: sampling.get_ode_sampler(sde, self, y=y[sl], **kwargs), y, minibatch)

    def enhance_batch(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="ald", N=50,
                      corrector_steps=1, snr=0.5, return_nfe=False, lengths=None, **kwargs):
        """B equal-length utterances y [B, L] in one sampler run.  Every op on the path is per utterance, the Langevin
        corrector runs with per-row step sizes and the ODE sampler with one Runge-Kutta step controller per row (the
        reference solves one utterance per solve_ivp call), so with INJECTED noise (noise_fn) row b equals enhance(y[b:b+1])
        for every sampler / predictor / corrector; with the in-kernel Philox stream (seed=) rows are independent draws but
        not the draws a batch-1 call with the same seed would make (the counter is the position in the batch).
        For sampler_type="ode" the returned nfe is the number of score evaluations executed (= the slowest row's count).
        lengths: ragged micro-batch - rows of different sample counts that share one padded frame count
        (storm_amd.distributed.bucket_by_frames); y is zero filled to the longest row and so is the result."""
        Y, peak, T_orig = self._prepare(y, lengths)
        if sampler_type == "pc":
            kwargs.setdefault("langevin_per_row", True)
            sampler = self.get_pc_sampler(predictor, corrector, Y, N=N, corrector_steps=corrector_steps, snr=snr,
                                          intermediate=False, **kwargs)
        elif sampler_type == "ode":
            kwargs.setdefault("per_row", True)       # one RK45 step controller per utterance (model.py:224: minibatch = 1)
            sampler = self.get_ode_sampler(Y, N=N, minibatch=None, **kwargs)
        else:
            raise ValueError("{} is not a valid sampler type!".format(sampler_type))
        sample, nfe = sampler()
        self.last_nfev_rows = getattr(sampler, "nfev_rows", None)     # ODE: the evaluations every row needed on its own
        x_hat = self.data_module.spec_to_wav(sample, T_orig, peak, lengths=lengths)
        return (x_hat, nfe) if return_nfe else x_hat

    def enhance(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="ald", N=50, corrector_steps=1,
                snr=0.5, timeit=False, scale_factor=None, return_stft=False, **kwargs):
        """One-call enhancement of one utterance y [1, L] (model.py:273-310)."""
        start = time.time()
        if return_stft:
            Y, peak, T_orig = self._prepare(y)
            sampler = self.get_pc_sampler(predictor, corrector, Y, N=N, corrector_steps=corrector_steps, snr=snr, **kwargs)
            sample, nfe = sampler()
            return sample.squeeze(), Y.squeeze(), T_orig, float(peak[0])
        x_hat, nfe = self.enhance_batch(y, sampler_type, predictor, corrector, N, corrector_steps, snr,
                                        return_nfe=True, **kwargs)
        x_hat = x_hat.squeeze().cpu()
        end = time.time()
        if timeit:
            rtf = (end - start) / (len(x_hat) / 16000)
            return x_hat, nfe, rtf
        return x_hat


class DiscriminativeModel(ScoreModel):
    """Predictive NCSN++ denoiser (model.py:320-370)."""

    def _set_score_sign(self):
        pass

    def forward(self, y):
        t = torch.ones(y.shape[0], device=y.device)
        return self.dnn(y, t)

    def enhance(self, y, **ignored_kwargs):
        with torch.no_grad():
            Y, peak, T_orig = self._prepare(y)
            X_hat = self(Y)
            return self.data_module.spec_to_wav(X_hat, T_orig, peak).squeeze()


class StochasticRegenerationModel(_Base):
    """StoRM: predictive denoiser followed by the score-based sampler around its output (model.py:392-780)."""

    def __init__(self, backbone_denoiser: str = "ncsnpp", backbone_score: str = "ncsnpp", sde: str = "ouve",
                 lr: float = 1e-4, ema_decay: float = 0.999, t_eps: float = 3e-2, nolog: bool = False,
                 num_eval_files: int = 50, loss_type_denoiser: str = "none", loss_type_score: str = "mse",
                 data_module_cls=None, mode="regen-joint-training", condition="both", **kwargs):
        super().__init__()
        kwargs_denoiser = kwargs                                           # same dict on purpose (model.py:416)
        kwargs_denoiser.update(input_channels=2)
        kwargs_denoiser.update(discriminative=True)
        self.denoiser_net = BackboneRegistry.get_by_name(backbone_denoiser)(**kwargs) if backbone_denoiser != "none" else None
        kwargs.update(input_channels=(6 if condition == "both" else 4))
        kwargs_denoiser.update(discriminative=False)
        self.score_net = BackboneRegistry.get_by_name(backbone_score)(**kwargs) if backbone_score != "none" else None
        if self.score_net is not None and hasattr(self.score_net, "negate_output"):
            self.score_net.negate_output = True
        self._init_common(sde, t_eps, ema_decay, data_module_cls, kwargs)
        self.ema = _EMA(self.parameters(), decay=ema_decay)
        self.condition, self.mode = condition, mode
        self.lr, self.num_eval_files, self.nolog = lr, num_eval_files, nolog

    def forward_score(self, x, t, score_conditioning, sde_input, **kwargs):
        """-score_net(cat[x] + conditioning, t)  (model.py:548-554)"""
        return self.score_net.forward_parts([x[:, 0]] + [c[:, 0] for c in score_conditioning], t)

    def forward_denoiser(self, y, **kwargs):
        return self.denoiser_net(y)

    def get_pc_sampler(self, predictor_name, corrector_name, y, N=None, minibatch=None, scale_factor=None,
                       conditioning=None, **kwargs):
        N = self.sde.N if N is None else N
        sde = self.sde.copy()
        sde.N = N
        kwargs = {"eps": self.t_eps, **kwargs}
        if minibatch is None:
            return sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self.forward_score, y=y,
                                           conditioning=conditioning, **kwargs)
        return self._sampler_minibatched(
            lambda sl: sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self.forward_score,
                                               y=y[sl], conditioning=[c[sl] for c in conditioning], **kwargs),
            y, minibatch)

    def enhance_batch(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="none", N=30,
                      corrector_steps=1, snr=0.5, denoiser_only=False, return_nfe=False, return_stft=False, lengths=None,
                      **kwargs):
        Y, peak, T_orig = self._prepare(y, lengths)
        kwargs.setdefault("langevin_per_row", True)
        nfe = 0
        with torch.no_grad():
            Y_denoised = self.forward_denoiser(Y) if self.denoiser_net is not None else None
            if self.score_net is not None and not denoiser_only:
                if self.condition == "noisy":
                    score_conditioning = [Y]
                elif self.condition == "post_denoiser":
                    score_conditioning = [Y_denoised]
                elif self.condition == "both":
                    score_conditioning = [Y, Y_denoised]
                else:
                    raise NotImplementedError(f"Don't know the conditioning you have wished for: {self.condition}")
                if sampler_type != "pc":
                    raise NotImplementedError("StoRM supports the PC sampler only (the reference's ODE path drops the "
                                              "conditioning, model.py:671-691)")
                sampler = self.get_pc_sampler(predictor, corrector, Y_denoised, N=N, corrector_steps=corrector_steps,
                                              snr=snr, intermediate=False, conditioning=score_conditioning, **kwargs)
                sample, nfe = sampler()
            else:
                sample = Y_denoised
        if return_stft:                                     # (model.py:766-767; a batch gets every row's normalisation factor)
            norm = float(peak[0]) if sample.shape[0] == 1 else peak.detach().reshape(-1).cpu()
            return sample.squeeze(), Y.squeeze(), T_orig, norm
        x_hat = self.data_module.spec_to_wav(sample, T_orig, peak, lengths=lengths)
        return (x_hat, nfe) if return_nfe else x_hat

    def enhance(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="none", N=30, corrector_steps=1,
                snr=0.5, timeit=False, scale_factor=None, return_stft=False, denoiser_only=False, **kwargs):
        """model.py:720-780; return_stft=True returns (sample, Y, T_orig, norm_factor) like the reference."""
        start = time.time()
        out = self.enhance_batch(y, sampler_type, predictor, corrector, N, corrector_steps, snr,
                                 denoiser_only=denoiser_only, return_nfe=True, return_stft=return_stft, **kwargs)
        if return_stft:
            return out
        x_hat, nfe = out
        x_hat = x_hat.squeeze().cpu()
        end = time.time()
        if timeit:
            return x_hat, nfe, (end - start) / (len(x_hat) / 16000)
        return x_hat
