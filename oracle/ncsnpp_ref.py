"""Oracle: NCSN++ score-network forward, restated functionally on PyTorch CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows the reference
``NCSNpp.forward`` (sgmse/backbones/ncsnpp.py:281-450) and its layers
(sgmse/backbones/ncsnpp_utils/layerspp.py, layers.py, up_or_down_sampling.py)
for the configuration family the hot path uses: ``resblock_type='biggan'``,
``fir=True`` with kernel [1,3,3,1], ``progressive='output_skip'``,
``progressive_input='input_skip'`` with ``sum`` combine, ``skip_rescale=True``,
Gaussian-Fourier embedding, ``centered=False``, ``spatial_channels=1``.

The network is addressed purely through a ``state_dict`` with the reference's
key names (``all_modules.<i>.<Layer>.<param>``, ``output_layer.*``), so the
same weights drive the reference, this oracle and the HIP engine.
"""
import math
from dataclasses import dataclass, field
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class NCSNppConfig:
    """Hyper-parameters that change the graph (ncsnpp.py:40-65)."""
    nf: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 2, 2)
    num_res_blocks: int = 1
    attn_resolutions: Tuple[int, ...] = (0,)
    image_size: int = 256
    input_channels: int = 4          # real channels = 2 x complex inputs
    discriminative: bool = False     # ncsnpp.py:80-86
    fourier_scale: float = 16.0

    @property
    def conditional(self):
        return not self.discriminative

    @property
    def scale_by_sigma(self):
        return not self.discriminative

    @property
    def in_ch(self):
        return 2 if self.discriminative else self.input_channels


NAMED_CONFIGS = {
    # ncsnpp.py:36-65, 460-470, 479-509
    "ncsnpp": dict(nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1, attn_resolutions=(0,)),
    "ncsnpplarge": dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,)),
    "ncsnpp12M": dict(nf=96, ch_mult=(1, 2, 2, 1), num_res_blocks=1, attn_resolutions=(0,)),
    "ncsnpp6M": dict(nf=96, ch_mult=(1, 1, 1, 1), num_res_blocks=1, attn_resolutions=(0,)),
}


def silu(x):
    return x * torch.sigmoid(x)


def group_norm(x, w, b):
    """nn.GroupNorm(min(C//4,32), C, eps=1e-6)  (layerspp.py:219, ncsnpp.py:238)."""
    C = x.shape[1]
    return F.group_norm(x, min(C // 4, 32), w, b, eps=1e-6)


def fir_up2(x):
    """upsample_2d(x, [1,3,3,1], factor=2)  (up_or_down_sampling.py:195-224).

    Zero-insert x2, zero boundary, separable taps [1,3,3,1]/8 per axis with
    gain 2 per axis: out[2i] = 3/4 x[i] + 1/4 x[i-1], out[2i+1] = 3/4 x[i] + 1/4 x[i+1].
    """
    def up_axis(v, dim):
        n = v.shape[dim]
        zero = torch.zeros_like(v.narrow(dim, 0, 1))
        prev = torch.cat([zero, v.narrow(dim, 0, n - 1)], dim)
        nxt = torch.cat([v.narrow(dim, 1, n - 1), zero], dim)
        even = 0.75 * v + 0.25 * prev
        odd = 0.75 * v + 0.25 * nxt
        out = torch.stack([even, odd], dim + 1)
        shape = list(v.shape)
        shape[dim] = 2 * n
        return out.reshape(shape)
    return up_axis(up_axis(x, 2), 3)


def fir_down2(x):
    """downsample_2d(x, [1,3,3,1], factor=2)  (up_or_down_sampling.py:227-257).

    out[o] = (x[2o-1] + 3 x[2o] + 3 x[2o+1] + x[2o+2]) / 8 per axis, zero boundary.
    """
    def down_axis(v, dim):
        n = v.shape[dim]
        pad = [0, 0] * (v.dim() - 1 - dim) + [1, 1]
        vp = F.pad(v, pad)
        idx = torch.arange(0, n, 2)
        a = vp.index_select(dim, idx)
        b = vp.index_select(dim, idx + 1)
        c = vp.index_select(dim, idx + 2)
        d = vp.index_select(dim, idx + 3)
        return (a + 3.0 * b + 3.0 * c + d) / 8.0
    return down_axis(down_axis(x, 2), 3)


def upfirdn2d(x, k, up_x=1, up_y=1, down_x=1, down_y=1, pad_x0=0, pad_x1=0, pad_y0=0, pad_y1=0):
    """The reference's native op for ANY parameter set (op/upfirdn2d.cpp:12-22; CPU form upfirdn2d_native, op/upfirdn2d.py:159-200;
    CUDA form op/upfirdn2d_kernel.cu:107-369), restated as a gather: x [N, H, W] planes, k [kh, kw] ->
        out[n, oy, ox] = sum_{ky,kx} k[kh-1-ky, kw-1-kx] * U[n, oy*down_y + ky - pad_y0, ox*down_x + kx - pad_x0]
    with U the zero-stuffed image (U[up_y*i, up_x*j] = x[i, j], zero elsewhere and outside), i.e. a true convolution (the reference
    flips the kernel for F.conv2d) of the padded / cropped zero-stuffed image followed by decimation;
    out_h = (H*up_y + pad_y0 + pad_y1 - kh) // down_y + 1 (likewise out_w)."""
    x, k = torch.as_tensor(x), torch.as_tensor(k)
    N, H, W = x.shape
    kh, kw = k.shape
    OH = (H * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    OW = (W * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    U = torch.zeros(N, H * up_y, W * up_x, dtype=x.dtype)
    U[:, ::up_y, ::up_x] = x
    out = torch.zeros(N, max(OH, 0), max(OW, 0), dtype=x.dtype)
    oy, ox = torch.arange(max(OH, 0)), torch.arange(max(OW, 0))
    for ky in range(kh):
        uy = oy * down_y + ky - pad_y0
        my = (uy >= 0) & (uy < H * up_y)
        for kx in range(kw):
            ux = ox * down_x + kx - pad_x0
            mx = (ux >= 0) & (ux < W * up_x)
            g = U[:, uy.clamp(0, H * up_y - 1)][:, :, ux.clamp(0, W * up_x - 1)]
            out += k[kh - 1 - ky, kw - 1 - kx] * g * (my[:, None] & mx[None, :]).to(x.dtype)
    return out


def nin(x, W, b):
    """layers.NIN (layers.py:548-557): y = x . W + b over the channel axis, W [Cin, Cout]."""
    return torch.einsum("bchw,cd->bdhw", x, W) + b[None, :, None, None]


class _SD:
    """Prefix view over a state_dict."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def __contains__(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return _SD(self.sd, self.prefix + p)


def resblock(m, x, temb, up=False, down=False):
    """ResnetBlockBigGANpp.forward (layerspp.py:242-274)."""
    h = silu(group_norm(x, m["GroupNorm_0.weight"], m["GroupNorm_0.bias"]))
    if up:
        h, x = fir_up2(h), fir_up2(x)
    elif down:
        h, x = fir_down2(h), fir_down2(x)
    h = F.conv2d(h, m["Conv_0.weight"], m["Conv_0.bias"], padding=1)
    if temb is not None:
        h = h + F.linear(silu(temb), m["Dense_0.weight"], m["Dense_0.bias"])[:, :, None, None]
    h = silu(group_norm(h, m["GroupNorm_1.weight"], m["GroupNorm_1.bias"]))
    h = F.conv2d(h, m["Conv_1.weight"], m["Conv_1.bias"], padding=1)
    if "Conv_2.weight" in m:
        x = F.conv2d(x, m["Conv_2.weight"], m["Conv_2.bias"])
    return (x + h) / math.sqrt(2.0)


def attnblock(m, x):
    """AttnBlockpp.forward (layerspp.py:75-91), skip_rescale=True."""
    B, C, H, W = x.shape
    h = group_norm(x, m["GroupNorm_0.weight"], m["GroupNorm_0.bias"])
    q = nin(h, m["NIN_0.W"], m["NIN_0.b"])
    k = nin(h, m["NIN_1.W"], m["NIN_1.b"])
    v = nin(h, m["NIN_2.W"], m["NIN_2.b"])
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(C) ** (-0.5))
    w = torch.reshape(w, (B, H, W, H * W))
    w = F.softmax(w, dim=-1)
    w = torch.reshape(w, (B, H, W, H, W))
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = nin(h, m["NIN_3.W"], m["NIN_3.b"])
    return (x + h) / math.sqrt(2.0)


def time_embedding(mods, cfg, t):
    """GaussianFourierProjection(log t) -> Linear -> SiLU -> Linear (ncsnpp.py:298-317, layerspp.py:39-41)."""
    x = torch.log(t)
    x_proj = x[:, None] * mods["0.W"][None, :] * 2 * math.pi
    temb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
    temb = F.linear(temb, mods["1.weight"], mods["1.bias"])
    temb = F.linear(silu(temb), mods["2.weight"], mods["2.bias"])
    return temb


def pack_complex(x):
    """complex [B,D,F,T] -> real [B,2D,F,T] as (re0, im0, re1, im1, ...) (ncsnpp.py:289-296, spatial_channels=1)."""
    parts = []
    for c in range(x.shape[1]):
        parts += [x[:, [c]].real, x[:, [c]].imag]
    return torch.cat(parts, dim=1)


def ncsnpp_forward(sd, cfg: NCSNppConfig, x, t=None, prefix=""):
    """NCSNpp.forward (ncsnpp.py:281-450).

    x: complex64 [B, in_ch/2, F, T];  t: float32 [B] (ignored when discriminative).
    Returns complex64 [B, 1, F, T]  (the raw network output; the score is its negative,
    sgmse/model.py:131-132).
    """
    root = _SD(sd, prefix)
    mods = root.sub("all_modules.")
    nres = len(cfg.ch_mult)
    x = pack_complex(x).to(torch.float32)
    midx = 1
    if cfg.conditional:
        temb = time_embedding(mods, cfg, t)
        midx = 3
    else:
        temb = None
    x = 2 * x - 1.0                                                  # ncsnpp.py:321-323
    ip = x
    hs = [F.conv2d(x, mods[f"{midx}.weight"], mods[f"{midx}.bias"], padding=1)]
    midx += 1
    for lvl in range(nres):
        for _ in range(cfg.num_res_blocks):
            h = resblock(mods.sub(f"{midx}."), hs[-1], temb)
            midx += 1
            if h.shape[-2] in cfg.attn_resolutions:                  # ncsnpp.py:338
                h = attnblock(mods.sub(f"{midx}."), h)
                midx += 1
            hs.append(h)
        if lvl != nres - 1:
            h = resblock(mods.sub(f"{midx}."), hs[-1], temb, down=True)
            midx += 1
            ip = fir_down2(ip)                                       # ncsnpp.py:353
            m = mods.sub(f"{midx}.")                                 # Combine, layerspp.py:52-57
            h = F.conv2d(ip, m["Conv_0.weight"], m["Conv_0.bias"]) + h
            midx += 1
            hs.append(h)
    h = hs[-1]
    h = resblock(mods.sub(f"{midx}."), h, temb); midx += 1
    h = attnblock(mods.sub(f"{midx}."), h); midx += 1
    h = resblock(mods.sub(f"{midx}."), h, temb); midx += 1
    pyramid = None
    for lvl in reversed(range(nres)):
        for _ in range(cfg.num_res_blocks + 1):
            h = resblock(mods.sub(f"{midx}."), torch.cat([h, hs.pop()], dim=1), temb)
            midx += 1
        if h.shape[-2] in cfg.attn_resolutions:                      # ncsnpp.py:385
            h = attnblock(mods.sub(f"{midx}."), h)
            midx += 1
        # output_skip pyramid (ncsnpp.py:389-410)
        ph = silu(group_norm(h, mods[f"{midx}.weight"], mods[f"{midx}.bias"]))
        midx += 1
        ph = F.conv2d(ph, mods[f"{midx}.weight"], mods[f"{midx}.bias"], padding=1)
        midx += 1
        pyramid = ph if pyramid is None else fir_up2(pyramid) + ph
        if lvl != 0:
            h = resblock(mods.sub(f"{midx}."), h, temb, up=True)
            midx += 1
    assert not hs
    h = pyramid
    if cfg.scale_by_sigma:
        h = h / t[:, None, None, None]                               # ncsnpp.py:441-443
    h = F.conv2d(h, root["output_layer.weight"], root["output_layer.bias"])
    return torch.complex(h[:, 0], h[:, 1]).unsqueeze(1)              # ncsnpp.py:446-449


def module_plan(cfg: NCSNppConfig):
    """Enumerates ``all_modules`` exactly as NCSNpp.__init__ does (ncsnpp.py:153-273).

    Returns a list of (index, kind, params) used to create seeded state_dicts
    without the reference.  kind in {gfp, linear, conv3, res, combine, attn, gn}.
    """
    nf, nres = cfg.nf, len(cfg.ch_mult)
    total = cfg.in_ch
    all_res = [cfg.image_size // (2 ** i) for i in range(nres)]
    plan = [("gfp", dict(n=nf))]
    if cfg.conditional:
        plan += [("linear", dict(i=2 * nf, o=4 * nf)), ("linear", dict(i=4 * nf, o=4 * nf))]
    plan.append(("conv3", dict(i=total, o=nf)))
    hs_c = [nf]
    in_ch = nf
    for lvl in range(nres):
        for _ in range(cfg.num_res_blocks):
            out_ch = nf * cfg.ch_mult[lvl]
            plan.append(("res", dict(i=in_ch, o=out_ch, resample=False)))
            in_ch = out_ch
            if all_res[lvl] in cfg.attn_resolutions:
                plan.append(("attn", dict(c=in_ch)))
            hs_c.append(in_ch)
        if lvl != nres - 1:
            plan.append(("res", dict(i=in_ch, o=in_ch, resample=True)))
            plan.append(("combine", dict(i=total, o=in_ch)))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    plan += [("res", dict(i=in_ch, o=in_ch, resample=False)), ("attn", dict(c=in_ch)),
             ("res", dict(i=in_ch, o=in_ch, resample=False))]
    for lvl in reversed(range(nres)):
        for _ in range(cfg.num_res_blocks + 1):
            out_ch = nf * cfg.ch_mult[lvl]
            plan.append(("res", dict(i=in_ch + hs_c.pop(), o=out_ch, resample=False)))
            in_ch = out_ch
        if all_res[lvl] in cfg.attn_resolutions:
            plan.append(("attn", dict(c=in_ch)))
        plan.append(("gn", dict(c=in_ch)))
        plan.append(("conv3", dict(i=in_ch, o=total)))
        if lvl != 0:
            plan.append(("res", dict(i=in_ch, o=in_ch, resample=True)))
    assert not hs_c
    return plan


def seeded_state_dict(cfg: NCSNppConfig, seed=0, prefix=""):
    """A full state_dict with the reference's key names and shapes,
    filled from a seeded CPU generator.  Values are *not* the reference's init: its
    ``init_scale=0`` layers would make outputs degenerate, so every tensor gets
    non-trivial values (weights ~ U(-a, a) with fan-avg scaling, biases ~ 0.1 N(0,1),
    norm gains ~ 1 + 0.1 N(0,1)).
    """
    g = torch.Generator().manual_seed(seed)

    def W(*shape):
        fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
        fan_out = shape[0] * (shape[2] * shape[3] if len(shape) == 4 else 1)
        a = math.sqrt(3.0 * 2.0 / (fan_in + fan_out))
        return (torch.rand(*shape, generator=g) * 2 - 1) * a

    def bias(n):
        return 0.1 * torch.randn(n, generator=g)

    def gain(n):
        return 1.0 + 0.1 * torch.randn(n, generator=g)

    sd = {}
    total = cfg.in_ch
    sd[prefix + "output_layer.weight"] = W(2, total, 1, 1)
    sd[prefix + "output_layer.bias"] = bias(2)
    for idx, (kind, p) in enumerate(module_plan(cfg)):
        k = f"{prefix}all_modules.{idx}."
        if kind == "gfp":
            sd[k + "W"] = torch.randn(p["n"], generator=g) * cfg.fourier_scale
        elif kind == "linear":
            sd[k + "weight"] = W(p["o"], p["i"]); sd[k + "bias"] = bias(p["o"])
        elif kind == "conv3":
            sd[k + "weight"] = W(p["o"], p["i"], 3, 3); sd[k + "bias"] = bias(p["o"])
        elif kind == "gn":
            sd[k + "weight"] = gain(p["c"]); sd[k + "bias"] = bias(p["c"])
        elif kind == "combine":
            sd[k + "Conv_0.weight"] = W(p["o"], p["i"], 1, 1); sd[k + "Conv_0.bias"] = bias(p["o"])
        elif kind == "attn":
            c = p["c"]
            sd[k + "GroupNorm_0.weight"] = gain(c); sd[k + "GroupNorm_0.bias"] = bias(c)
            for j in range(4):
                sd[k + f"NIN_{j}.W"] = W(c, c); sd[k + f"NIN_{j}.b"] = bias(c)
        elif kind == "res":
            i, o = p["i"], p["o"]
            sd[k + "GroupNorm_0.weight"] = gain(i); sd[k + "GroupNorm_0.bias"] = bias(i)
            sd[k + "Conv_0.weight"] = W(o, i, 3, 3); sd[k + "Conv_0.bias"] = bias(o)
            sd[k + "Dense_0.weight"] = W(o, 4 * cfg.nf); sd[k + "Dense_0.bias"] = bias(o)
            sd[k + "GroupNorm_1.weight"] = gain(o); sd[k + "GroupNorm_1.bias"] = bias(o)
            sd[k + "Conv_1.weight"] = W(o, o, 3, 3); sd[k + "Conv_1.bias"] = bias(o)
            if i != o or p["resample"]:
                sd[k + "Conv_2.weight"] = W(o, i, 1, 1); sd[k + "Conv_2.bias"] = bias(o)
    return sd
