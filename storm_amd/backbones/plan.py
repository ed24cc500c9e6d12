"""NCSN++ configuration and module enumeration (the reference ``NCSNpp.__init__``, sgmse/backbones/ncsnpp.py:153-273)
for the configuration family of the hot path: what the nn.Module shell needs to own the reference's state_dict.

The planner itself (parameter arena layout + fused op program + liveness-based workspace) lives behind the C ABI
(csrc/ncsnpp_graph.hip: storm_ncsnpp_create / storm_ncsnpp_forward); tests/py_planner.py keeps an independent Python
restatement of it that the tests compare op for op.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .. import _lib as L

# op codes (include/storm_hip.h)
OP_MEMSET, OP_PACK_INPUT, OP_TEMB, OP_DENSE, OP_CONV, OP_GN_STATS, OP_GN_APPLY, OP_FIR_UP, OP_FIR_DOWN, \
    OP_SOFTMAX, OP_OUTPUT_HEAD, OP_GN_FINALIZE, OP_ATTENTION, OP_INPUT_PYRAMID, OP_OUTPUT_PYRAMID = range(15)

# buffer slots of storm_program_run
BUF_WS, BUF_PARAMS, BUF_IN0, BUF_IN1, BUF_IN2, BUF_T, BUF_OUT = range(7)
N_BUFS = 7

ALIGN = 256


def _up(x, m):
    return (x + m - 1) // m * m


@dataclass(frozen=True)
class NCSNppConfig:
    """Graph-shaping hyper-parameters (ncsnpp.py:40-65)."""
    nf: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 2, 2)
    num_res_blocks: int = 1
    attn_resolutions: Tuple[int, ...] = (0,)
    image_size: int = 256
    input_channels: int = 4
    discriminative: bool = False
    fourier_scale: float = 16.0

    @property
    def conditional(self):
        return not self.discriminative

    @property
    def total_channels(self):
        return 2 if self.discriminative else self.input_channels


def module_list(cfg: NCSNppConfig):
    """``all_modules`` in registration order: list of (kind, params).  kind in
    {gfp, linear, conv3, res, combine, attn, gn}.  Same order as ncsnpp.py:153-273."""
    nf, nres, total = cfg.nf, len(cfg.ch_mult), cfg.total_channels
    all_res = [cfg.image_size // (2 ** i) for i in range(nres)]
    mods = [("gfp", dict(n=nf))]
    if cfg.conditional:
        mods += [("linear", dict(i=2 * nf, o=4 * nf)), ("linear", dict(i=4 * nf, o=4 * nf))]
    mods.append(("conv3", dict(i=total, o=nf)))
    hs_c, in_ch = [nf], nf
    for lvl in range(nres):
        for _ in range(cfg.num_res_blocks):
            out_ch = nf * cfg.ch_mult[lvl]
            mods.append(("res", dict(i=in_ch, o=out_ch, resample=False)))
            in_ch = out_ch
            if all_res[lvl] in cfg.attn_resolutions:
                mods.append(("attn", dict(c=in_ch)))
            hs_c.append(in_ch)
        if lvl != nres - 1:
            mods.append(("res", dict(i=in_ch, o=in_ch, resample=True)))
            mods.append(("combine", dict(i=total, o=in_ch)))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods += [("res", dict(i=in_ch, o=in_ch, resample=False)), ("attn", dict(c=in_ch)),
             ("res", dict(i=in_ch, o=in_ch, resample=False))]
    for lvl in reversed(range(nres)):
        for _ in range(cfg.num_res_blocks + 1):
            out_ch = nf * cfg.ch_mult[lvl]
            mods.append(("res", dict(i=in_ch + hs_c.pop(), o=out_ch, resample=False)))
            in_ch = out_ch
        if all_res[lvl] in cfg.attn_resolutions:
            mods.append(("attn", dict(c=in_ch)))
        mods.append(("gn", dict(c=in_ch)))
        mods.append(("conv3", dict(i=in_ch, o=total)))
        if lvl != 0:
            mods.append(("res", dict(i=in_ch, o=in_ch, resample=True)))
    assert not hs_c
    return mods


def state_dict_shapes(cfg: NCSNppConfig):
    """name -> shape of every tensor in the reference state_dict."""
    total = cfg.total_channels
    shapes = {"output_layer.weight": (2, total, 1, 1), "output_layer.bias": (2,)}
    for idx, (kind, p) in enumerate(module_list(cfg)):
        k = f"all_modules.{idx}."
        if kind == "gfp":
            shapes[k + "W"] = (p["n"],)
        elif kind == "linear":
            shapes[k + "weight"] = (p["o"], p["i"]); shapes[k + "bias"] = (p["o"],)
        elif kind == "conv3":
            shapes[k + "weight"] = (p["o"], p["i"], 3, 3); shapes[k + "bias"] = (p["o"],)
        elif kind == "gn":
            shapes[k + "weight"] = (p["c"],); shapes[k + "bias"] = (p["c"],)
        elif kind == "combine":
            shapes[k + "Conv_0.weight"] = (p["o"], p["i"], 1, 1); shapes[k + "Conv_0.bias"] = (p["o"],)
        elif kind == "attn":
            c = p["c"]
            shapes[k + "GroupNorm_0.weight"] = (c,); shapes[k + "GroupNorm_0.bias"] = (c,)
            for j in range(4):
                shapes[k + f"NIN_{j}.W"] = (c, c); shapes[k + f"NIN_{j}.b"] = (c,)
        elif kind == "res":
            i, o = p["i"], p["o"]
            shapes[k + "GroupNorm_0.weight"] = (i,); shapes[k + "GroupNorm_0.bias"] = (i,)
            shapes[k + "Conv_0.weight"] = (o, i, 3, 3); shapes[k + "Conv_0.bias"] = (o,)
            shapes[k + "Dense_0.weight"] = (o, 4 * cfg.nf); shapes[k + "Dense_0.bias"] = (o,)
            shapes[k + "GroupNorm_1.weight"] = (o,); shapes[k + "GroupNorm_1.bias"] = (o,)
            shapes[k + "Conv_1.weight"] = (o, o, 3, 3); shapes[k + "Conv_1.bias"] = (o,)
            if i != o or p["resample"]:
                shapes[k + "Conv_2.weight"] = (o, i, 1, 1); shapes[k + "Conv_2.bias"] = (o,)
    return shapes
