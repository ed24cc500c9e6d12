"""Per-op parity: every C-ABI entry point against the oracle / plain torch fp32 on the same
seeded inputs.  Runs on the host simulation (CPU suite) and on the MI355X (-m gpu)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import frontend_ref as FR
from oracle import ncsnpp_ref as NR
from oracle import sde_ref as SR
from tests.backend import dev, switch, nchw, nhwc, tol  # noqa: F401
from tests.util import rel_l2

DTYPES = [torch.float32, torch.bfloat16]


def q(x, dtype):
    """quantise a reference input the way the engine sees it"""
    return x.to(dtype).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 24, 10, 40, 3), (1, 72, 136, 9, 33, 3), (2, 8, 4, 8, 32, 3),
                                   (1, 40, 40, 5, 7, 1), (1, 128, 128, 16, 64, 3)])
def test_conv(dev, dtype, shape):
    from storm_amd import ops
    B, Cin, Cout, H, W, k = shape
    if dev.type == "cpu" and Cin * Cout > 72 * 136:
        pytest.skip("large case: GPU only")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    xq = nhwc(x).to(dtype).to(dev)
    wp = ops.pack_conv_weight(w.to(dev), dtype)
    y = ops.conv([ops.Seg(xq, wp, k * k)], Cout, bias=b.to(dev)).float().cpu()
    ref = F.conv2d(q(x, dtype), q(w, dtype), b, padding=k // 2)
    assert rel_l2(nchw(y)[:, :Cout], ref) < tol(dtype, 2e-6, 6e-3)
    if y.shape[-1] > Cout:
        assert float(y[..., Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_persistent_tile_loop(dev, dtype, k, switch):
    """More tiles than resident workgroups: every workgroup walks several tiles (output + statistics partials)."""
    from storm_amd import ops
    switch("STORM_CONV_CUS", 8)
    g = torch.Generator().manual_seed(21)
    B, Cin, Cout, H, W = 3, 16, 40, 35, 70
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    xq = nhwc(x).to(dtype).to(dev)
    y, part = ops.conv([ops.Seg(xq, ops.pack_conv_weight(w.to(dev), dtype), k * k)], Cout, bias=b.to(dev), gn_partials=True)
    ref = F.conv2d(q(x, dtype), q(w, dtype), b, padding=k // 2)
    assert rel_l2(nchw(y.float().cpu())[:, :Cout], ref) < tol(dtype, 2e-6, 6e-3)
    rtol = 1e-5 if dtype == torch.float32 else 2e-3
    st, sref = ops.gn_finalize(part).cpu(), ops.gn_stats(y).cpu()
    assert torch.allclose(st, sref, rtol=rtol, atol=rtol * float(sref.abs().max()))


@pytest.mark.parametrize("variant", [3, 7, 9])
@pytest.mark.parametrize("case", ["plain", "block_tail", "gn_fused", "ragged", "deep_k", "skip", "plain@8", "block_tail@8", "gn_fused@8",
                                  "skip@8", "deep_k@8", "plain:f16", "gn_fused:f16", "deep_k:f16"])
def test_conv_pipelined_kernels(dev, variant, case, switch):
    """conv_pipe.hip (chunk-unrolled LDS-DMA pipeline, 256-cout tile; variant 9: its 128-cout tile for layers with few pixel
    tiles) on shapes the default dispatch would give to conv_igemm.hip: plain 3x3, fused 1x1 shortcut over a concat, fused GroupNorm-apply operand, ragged sizes, many
    K-chunks; "@8": as if the device had 8 CUs, so every persistent workgroup walks several tiles (next-tile prefetch
    before the epilogue, staging beside the landing loads)."""
    from storm_amd import ops
    switch("STORM_CONV_VARIANT", variant)
    if case.endswith("@8"):
        switch("STORM_CONV_CUS", 8)
        case = case[:-2]
    dtype = torch.bfloat16
    if case.endswith(":f16"):                               # fp16 operands (v_mfma_f32_32x32x16_f16): same kernel template
        dtype, case = torch.float16, case[:-4]
    g = torch.Generator().manual_seed(31)
    if case == "plain":
        B, Cin, Cout, H, W = 2, 72, 288, 19, 45
        x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
        b = torch.randn(Cout, generator=g)
        y, part = ops.conv([ops.Seg(nhwc(x).to(dtype).to(dev), ops.pack_conv_weight(w.to(dev), dtype), 9)], Cout,
                           bias=b.to(dev), gn_partials=True)
        ref = F.conv2d(q(x, dtype), q(w, dtype), b, padding=1)
        assert rel_l2(nchw(y.float().cpu())[:, :Cout], ref) < 6e-3
        st, sref = ops.gn_finalize(part).cpu(), ops.gn_stats(y).cpu()
        assert torch.allclose(st, sref, rtol=2e-3, atol=2e-3 * float(sref.abs().max()))
    elif case == "skip":                               # identity shortcut of a residual block: skip operands are fetched a pass ahead
        B, Cin, Cout, H, W = 2, 72, 288, 19, 45        # (ragged rows / columns / couts: fetches past the image are masked)
        x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
        b = torch.randn(Cout, generator=g)
        sk = torch.randn(B, Cout, H, W, generator=g)
        y, part = ops.conv([ops.Seg(nhwc(x).to(dtype).to(dev), ops.pack_conv_weight(w.to(dev), dtype), 9)], Cout,
                           bias=b.to(dev), skip=nhwc(sk).to(dtype).to(dev), scale=2 ** -0.5, gn_partials=True)
        ref = (F.conv2d(q(x, dtype), q(w, dtype), b, padding=1) + q(sk, dtype)) * 2 ** -0.5
        assert rel_l2(nchw(y.float().cpu())[:, :Cout], ref) < 6e-3
        st, sref = ops.gn_finalize(part).cpu(), ops.gn_stats(y).cpu()
        assert torch.allclose(st, sref, rtol=2e-3, atol=2e-3 * float(sref.abs().max()))
    elif case == "block_tail":
        B, H, W, Ca, Cb, Co = 2, 9, 35, 72, 16, 136
        h = torch.randn(B, Co, H, W, generator=g)
        xa, xb = torch.randn(B, Ca, H, W, generator=g), torch.randn(B, Cb, H, W, generator=g)
        w1 = torch.randn(Co, Co, 3, 3, generator=g) * 0.05; w2 = torch.randn(Co, Ca + Cb, 1, 1, generator=g) * 0.2
        bias, tb = torch.randn(Co, generator=g), torch.randn(B, 200, generator=g)
        segs = [ops.Seg(nhwc(h).to(dtype).to(dev), ops.pack_conv_weight(w1.to(dev), dtype), 9),
                ops.Seg(nhwc(xa).to(dtype).to(dev), ops.pack_conv_weight(w2.to(dev), dtype), 1, src_b=nhwc(xb).to(dtype).to(dev))]
        tbd = tb.to(dev)
        y = ops.conv(segs, Co, bias=bias.to(dev), tbias=tbd[:, 8:], scale=1 / math.sqrt(2)).float().cpu()
        ref = (F.conv2d(q(h, dtype), q(w1, dtype), padding=1) + F.conv2d(q(torch.cat([xa, xb], 1), dtype), q(w2, dtype))
               + bias[None, :, None, None] + tb[:, 8:8 + Co, None, None]) / math.sqrt(2)
        assert rel_l2(nchw(y), ref) < 6e-3
    elif case == "gn_fused":
        B, C0, Ca, Cb, Co, H, W = 2, 8, 72, 56, 40, 10, 36
        x0 = torch.randn(B, C0, H, W, generator=g)
        wa, wb = torch.randn(Ca, C0, 3, 3, generator=g) * 0.4, torch.randn(Cb, C0, 1, 1, generator=g) * 0.7
        w = torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.1
        gam, bet = 1 + 0.1 * torch.randn(Ca + Cb, generator=g), 0.1 * torch.randn(Ca + Cb, generator=g)
        x0d = nhwc(x0).to(dtype).to(dev)
        xa, pa = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wa.to(dev), dtype), 9)], Ca, gn_partials=True)
        xb, pb = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wb.to(dev), dtype), 1)], Cb, gn_partials=True)
        st, ss = ops.gn_finalize(pa, pb, gamma=gam.to(dev), beta=bet.to(dev), count=H * W)
        y = ops.conv([ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, src_b=xb, gn_ss=ss, gn_silu=True)], Co)
        xcat = torch.cat([nchw(xa.float().cpu()), nchw(xb.float().cpu())], 1)
        a = NR.silu(NR.group_norm(xcat, gam, bet))
        ref = F.conv2d(q(a, dtype), q(w, dtype), padding=1)
        assert rel_l2(nchw(y.float().cpu()), ref) < 1e-2
    elif case == "deep_k":
        # many chunks: 3 + 2 nine-tap chunks over a concat with a fused GroupNorm (last chunk of each run ragged), then 3 + 2
        # one-tap chunks of the fused shortcut: every ring slot / patch-buffer parity / weight-run change is exercised
        B, C0, Ca, Cb, Sa, Sb, Co, H, W = 1, 8, 136, 88, 136, 72, 264, 9, 33
        x0 = torch.randn(B, C0, H, W, generator=g)
        wa, wb = torch.randn(Ca, C0, 3, 3, generator=g) * 0.4, torch.randn(Cb, C0, 1, 1, generator=g) * 0.7
        w = torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.05
        sa, sb = torch.randn(B, Sa, H, W, generator=g), torch.randn(B, Sb, H, W, generator=g)
        w2 = torch.randn(Co, Sa + Sb, 1, 1, generator=g) * 0.1
        bias = torch.randn(Co, generator=g)
        gam, bet = 1 + 0.1 * torch.randn(Ca + Cb, generator=g), 0.1 * torch.randn(Ca + Cb, generator=g)
        x0d = nhwc(x0).to(dtype).to(dev)
        xa, pa = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wa.to(dev), dtype), 9)], Ca, gn_partials=True)
        xb, pb = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wb.to(dev), dtype), 1)], Cb, gn_partials=True)
        st, ss = ops.gn_finalize(pa, pb, gamma=gam.to(dev), beta=bet.to(dev), count=H * W)
        y = ops.conv([ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, src_b=xb, gn_ss=ss, gn_silu=True),
                      ops.Seg(nhwc(sa).to(dtype).to(dev), ops.pack_conv_weight(w2.to(dev), dtype), 1, src_b=nhwc(sb).to(dtype).to(dev))],
                     Co, bias=bias.to(dev), scale=0.5)
        xcat = torch.cat([nchw(xa.float().cpu()), nchw(xb.float().cpu())], 1)
        a = NR.silu(NR.group_norm(xcat, gam, bet))
        ref = (F.conv2d(q(a, dtype), q(w, dtype), padding=1) + F.conv2d(q(torch.cat([sa, sb], 1), dtype), q(w2, dtype))
               + bias[None, :, None, None]) * 0.5
        assert rel_l2(nchw(y.float().cpu())[:, :Co], ref) < 1e-2
    else:
        B, Cin, Cout, H, W = 1, 24, 8 * 5, 5, 7                       # less than one pixel tile, Cout far below the tile
        x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
        y = ops.conv([ops.Seg(nhwc(x).to(dtype).to(dev), ops.pack_conv_weight(w.to(dev), dtype), 9)], Cout).float().cpu()
        assert rel_l2(nchw(y)[:, :Cout], F.conv2d(q(x, dtype), q(w, dtype), padding=1)) < 6e-3


@pytest.mark.parametrize("cus", [0, 8])
def test_conv_pipe_half_tile_equals_full_tile_bit_for_bit(dev, cus, switch):
    """conv_pipe_kernel<T, 128, 8> (variant 9: 128 couts per workgroup, a wave owns 64 couts x 64 pixels) walks the same chunk
    descriptors in the same order as <T, 256, 8>: the output is bit-identical (the GroupNorm partials to fp32 rounding) - 2 + 1 nine-tap chunks over a
    concat with a fused GroupNorm operand, 2 one-tap chunks of a fused shortcut, ragged rows / columns / couts, with and without
    the persistent tile walk."""
    from storm_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    B, C0, Ca, Cb, Sa, Co, H, W = 2, 8, 104, 56, 72, 264, 11, 37
    x0 = torch.randn(B, C0, H, W, generator=g)
    wa, wb = torch.randn(Ca, C0, 3, 3, generator=g) * 0.4, torch.randn(Cb, C0, 1, 1, generator=g) * 0.7
    w = torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.05
    sa = torch.randn(B, Sa, H, W, generator=g)
    w2 = torch.randn(Co, Sa, 1, 1, generator=g) * 0.1
    bias, tb = torch.randn(Co, generator=g), torch.randn(B, Co, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(Ca + Cb, generator=g), 0.1 * torch.randn(Ca + Cb, generator=g)
    x0d = nhwc(x0).to(dtype).to(dev)
    xa, pa = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wa.to(dev), dtype), 9)], Ca, gn_partials=True)
    xb, pb = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wb.to(dev), dtype), 1)], Cb, gn_partials=True)
    _, ss = ops.gn_finalize(pa, pb, gamma=gam.to(dev), beta=bet.to(dev), count=H * W)
    segs = [ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, src_b=xb, gn_ss=ss, gn_silu=True),
            ops.Seg(nhwc(sa).to(dtype).to(dev), ops.pack_conv_weight(w2.to(dev), dtype), 1)]
    kw = dict(bias=bias.to(dev), tbias=tb.to(dev), scale=0.5)
    if cus:
        switch("STORM_CONV_CUS", cus)
    out = {}
    for variant in (3, 9):
        switch("STORM_CONV_VARIANT", variant)
        assert ops.conv_kernel_name(segs, Co, **kw).endswith({3: "256, 8, 0>", 9: "128, 8, 0>"}[variant])
        out[variant] = ops.conv(segs, Co, gn_partials=True, **kw)
    assert torch.equal(out[3][0], out[9][0])
    # (the statistics partials of an 8-row tile are summed over 4 wave rows of 2 pixel rows instead of 2 x 4: fp32 rounding apart)
    assert torch.allclose(out[3][1], out[9][1], rtol=2e-5, atol=2e-5 * float(out[3][1].abs().max()))



@pytest.mark.parametrize("cus", [0, 3])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_group_equals_its_problems_own_launches_bit_for_bit(dev, dtype, cus, switch):
    """storm_conv_group (conv_pipe.hip, GROUP instantiation): ONE launch over the pixel tiles of three problems of one layer - the ragged
    micro-batches of a stream (BASELINE.json configs[4]): different batch sizes and widths (ragged tile rows / columns), own tensors, the
    same weights; 2 + 1 nine-tap chunks over a concat with a fused GroupNorm operand + 2 one-tap chunks of a fused shortcut, per-row
    temb bias, skip operand, GroupNorm partials.  Every problem's output equals its OWN storm_conv launch bit for bit with both tiles (256 /
    128 couts per workgroup; the partials to fp32 rounding for the other tile), with and without the persistent walk (3 pretend CUs: a
    workgroup's consecutive tiles belong to different problems)."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(78)
    C0, Ca, Cb, Sa, Co = 8, 104, 56, 72, 264
    wa, wb = torch.randn(Ca, C0, 3, 3, generator=g) * 0.4, torch.randn(Cb, C0, 1, 1, generator=g) * 0.7
    w = ops.pack_conv_weight((torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.05).to(dev), dtype)
    w2 = ops.pack_conv_weight((torch.randn(Co, Sa, 1, 1, generator=g) * 0.1).to(dev), dtype)
    wa, wb = ops.pack_conv_weight(wa.to(dev), dtype), ops.pack_conv_weight(wb.to(dev), dtype)
    bias = torch.randn(Co, generator=g).to(dev)
    gam, bet = (1 + 0.1 * torch.randn(Ca + Cb, generator=g)).to(dev), (0.1 * torch.randn(Ca + Cb, generator=g)).to(dev)
    problems = []
    for B, H, W in ((2, 11, 37), (1, 11, 70), (3, 11, 5)):
        x0d = nhwc(torch.randn(B, C0, H, W, generator=g)).to(dtype).to(dev)
        xa, pa = ops.conv([ops.Seg(x0d, wa, 9)], Ca, gn_partials=True)
        xb, pb = ops.conv([ops.Seg(x0d, wb, 1)], Cb, gn_partials=True)
        _, ss = ops.gn_finalize(pa, pb, gamma=gam, beta=bet, count=H * W)
        segs = [ops.Seg(xa, w, 9, src_b=xb, gn_ss=ss, gn_silu=True), ops.Seg(nhwc(torch.randn(B, Sa, H, W, generator=g)).to(dtype).to(dev), w2, 1)]
        skip = nhwc(torch.randn(B, Co, H, W, generator=g)).to(dtype).to(dev)
        problems.append((segs, dict(bias=bias, tbias=torch.randn(B, Co, generator=g).to(dev), skip=skip, scale=0.5)))
    if cus:
        switch("STORM_CONV_CUS", cus)
    own = {}
    for variant in (3, 9):
        switch("STORM_CONV_VARIANT", variant)
        own[variant] = [ops.conv(segs, Co, gn_partials=True, **kw) for segs, kw in problems]
    switch("STORM_CONV_VARIANT", -1)
    for bn, variant in ((256, 3), (128, 9)):
        outs, parts = ops.conv_group(problems, Co, gn_partials=True, bn=bn)
        for p in range(3):
            assert torch.equal(outs[p], own[variant][p][0]), (bn, p)
            assert torch.equal(outs[p], own[3][p][0])                          # (the two tiles write the same bits, test above)
            assert torch.equal(parts[p], own[variant][p][1]), (bn, p, "partials")
    # not one layer (another weight tensor in problem 1): refused, the caller runs them one by one
    other = ops.pack_conv_weight((torch.randn(Co, Sa, 1, 1, generator=g) * 0.1).to(dev), dtype)
    bad = [problems[0], ([problems[1][0][0], ops.Seg(problems[1][0][1].src_a, other, 1)], problems[1][1])]
    with pytest.raises(Exception):
        ops.conv_group(bad, Co)


@pytest.mark.parametrize("C,gn,cus", [(128, True, 0), (256, True, 3), (128, False, 0)])
def test_conv_group_narrow_output_equals_own_launches(dev, C, gn, cus, switch):
    """The grouped form of conv_narrow.hip (the output pyramid's conv3x3(act(GroupNorm(h))) -> 4 planes, ncsnpp.py:389-410): three problems of one
    layer - different batch sizes and image sizes (tiles cut by both edges), own tensors and (scale, shift) tables - in ONE launch; every
    problem equals its own storm_conv launch bit for bit, with and without the persistent walk across problems."""
    from storm_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(79)
    w = ops.pack_conv_weight((torch.randn(4, C, 3, 3, generator=g) * 0.05).to(dev), dtype)
    bias = torch.randn(4, generator=g).to(dev)
    problems = []
    for B, H, W in ((2, 19, 45), (1, 21, 70), (3, 9, 33)):
        x = nhwc(torch.randn(B, C, H, W, generator=g)).to(dtype).to(dev)
        ss = ops.pack_gn_ss(1 + 0.1 * torch.randn(B, C, generator=g), 0.1 * torch.randn(B, C, generator=g)).to(dev) if gn else None
        problems.append(([ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)], dict(bias=bias, outC=8)))
    if cus:
        switch("STORM_CONV_CUS", cus)
    own = [ops.conv(segs, 4, **kw) for segs, kw in problems]
    assert "conv_narrow" in ops.conv_kernel_name(problems[0][0], 4, **problems[0][1])
    outs = ops.conv_group(problems, 4)
    for p in range(3):
        assert torch.equal(outs[p], own[p]), p


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["c128_gn", "c256_gn_walk", "c128_plain", "c256_two_planes"])
def test_conv_narrow_output_kernel(dev, dtype, case, switch):
    """conv_narrow.hip - the 3x3 convolutions of the output pyramid (ncsnpp.py:389-410: conv3x3(act(GroupNorm(h))) -> 4 planes; 2 in
    the discriminative net): one 36-row 1x1 GEMM over the haloed region + a nine-point gather.  Against F.conv2d and the generic
    kernel; ragged image sizes (tiles cut by both image edges), several tiles and batch items per persistent workgroup ("walk": as if
    the device had 3 CUs - the affine table is reloaded when the batch item changes), plain operand, missing output planes zero."""
    from storm_amd import ops
    C, B, H, W, Cout, gn, cus = {"c128_gn": (128, 2, 19, 45, 4, True, 0), "c256_gn_walk": (256, 3, 21, 37, 4, True, 3),
                                 "c128_plain": (128, 1, 16, 32, 4, False, 0), "c256_two_planes": (256, 2, 9, 33, 2, True, 0)}[case]
    g = torch.Generator().manual_seed(91)
    dd = lambda t: t.to(dtype).to(dev)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05
    bias = torch.randn(Cout, generator=g)
    xa = dd(nhwc(x))
    ss, a_ref = None, q(x, dtype)
    if gn:
        gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        st = ops.gn_stats(xa).cpu()
        n = (C // 32) * H * W
        mean = st[:, :, 0] / n
        rstd = 1.0 / torch.sqrt(st[:, :, 1] / n - mean * mean + 1e-6)
        scale = (rstd.repeat_interleave(C // 32, 1) * gam[None].double()).float()
        shift = (bet[None].double() - mean.repeat_interleave(C // 32, 1) * scale.double()).float()
        ss = ops.pack_gn_ss(scale, shift).to(dev)
        a_ref = NR.silu(q(x, dtype) * scale[:, :, None, None] + shift[:, :, None, None])
    segs = [ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, gn_ss=ss, gn_silu=True)]
    kw = dict(bias=bias.to(dev), outC=8)
    switch("STORM_CONV_VARIANT", 0)                               # the generic 32-cout tile
    y_generic = ops.conv(segs, Cout, **kw)
    assert ops.conv_kernel_name(segs, Cout, **kw).startswith("storm::conv_igemm_kernel")
    switch("STORM_CONV_VARIANT", -1)
    if cus:
        switch("STORM_CONV_CUS", cus)
    assert ops.conv_kernel_name(segs, Cout, **kw).startswith("storm::conv_narrow_kernel")
    y = ops.conv(segs, Cout, **kw)
    yc = nchw(y.float().cpu())
    ref = F.conv2d(q(a_ref, dtype), q(w, dtype), bias, padding=1)
    assert rel_l2(yc[:, :Cout], ref) < 6e-3
    assert float(yc[:, Cout:].abs().max()) == 0.0
    assert rel_l2(yc, nchw(y_generic.float().cpu())) < 4e-3


@pytest.mark.parametrize("shape", [(1, 1, 1, 128, 4), (2, 3, 5, 128, 1), (1, 2, 70, 256, 3), (1, 41, 3, 128, 4)])
def test_conv_narrow_output_kernel_degenerate_images(dev, shape):
    """conv_narrow.hip on images smaller than its 20 x 32-pixel tile in one or both directions (one pixel; one row band; one column
    band), 1 / 3 / 4 output planes: every region fragment is mostly padding."""
    from storm_amd import ops
    B, H, W, C, Cout = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    segs = [ops.Seg(nhwc(x).to(torch.bfloat16).to(dev), ops.pack_conv_weight(w.to(dev), torch.bfloat16), 9)]
    assert ops.conv_kernel_name(segs, Cout, bias=b.to(dev), outC=8).startswith("storm::conv_narrow_kernel")
    y = nchw(ops.conv(segs, Cout, bias=b.to(dev), outC=8).float().cpu())
    assert rel_l2(y[:, :Cout], F.conv2d(q(x, torch.bfloat16), q(w, torch.bfloat16), b, padding=1)) < 6e-3
    assert float(y[:, Cout:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["stem", "stem_ragged", "combine", "combine_ragged"])
def test_conv_thin_input_kernel(dev, dtype, case, switch):
    """conv_thin.hip - the convolutions over an 8-channel input (stem conv3x3 of ncsnpp.py:183, input-skip conv1x1 + h of
    layerspp.py:44-59): operands straight from global memory.  Against F.conv2d and against the generic kernel (same statistics-
    partial layout), ragged image sizes, 4 of the 8 input channels in use, 128 / 256 / 96 output channels."""
    from storm_amd import ops
    B, H, W, Cout, k, with_skip = {"stem": (2, 16, 64, 128, 3, False), "stem_ragged": (2, 19, 45, 96, 3, False),
                                   "combine": (2, 8, 64, 256, 1, True), "combine_ragged": (1, 13, 37, 128, 1, True)}[case]
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 8, H, W, generator=g)
    x[:, 4:] = 0.0                                          # pack_input leaves channels 4-7 (6-7) zero
    w = torch.randn(Cout, 8, k, k, generator=g) * 0.3
    bias = torch.randn(Cout, generator=g)
    outC = ops.round_up(Cout, 8)
    skip = torch.randn(B, outC, H, W, generator=g) if with_skip else None
    dd = lambda t: t.to(dtype).to(dev)
    segs = [ops.Seg(dd(nhwc(x)), ops.pack_conv_weight(w.to(dev), dtype), k * k)]
    kw = dict(bias=bias.to(dev), skip=dd(nhwc(skip)) if with_skip else None, scale=0.5 if with_skip else 1.0)
    assert ops.conv_kernel_name(segs, Cout, **kw).startswith("storm::conv_thin_kernel")
    y, part = ops.conv(segs, Cout, gn_partials=True, **kw)
    ref = F.conv2d(q(x, dtype), q(w, dtype), bias, padding=k // 2)
    if with_skip:
        ref = (ref + q(skip, dtype)[:, :Cout]) * 0.5
    yc = nchw(y.float().cpu())
    assert rel_l2(yc[:, :Cout], ref) < (6e-3 if dtype == torch.bfloat16 else 1e-3)
    switch("STORM_CONV_VARIANT", 0)
    assert ops.conv_kernel_name(segs, Cout, **kw).startswith("storm::conv_igemm_kernel")
    y0, part0 = ops.conv(segs, Cout, gn_partials=True, **kw)
    assert rel_l2(yc, nchw(y0.float().cpu())) < 3e-3 and part.shape == part0.shape
    assert torch.allclose(part.cpu(), part0.cpu(), rtol=2e-2, atol=2e-2 * float(part0.abs().max()))
    st, sref = ops.gn_finalize(part).cpu(), ops.gn_stats(y).cpu()
    assert torch.allclose(st, sref, rtol=2e-3, atol=2e-3 * float(sref.abs().max()))


@pytest.mark.parametrize("k,with_skip", [(3, False), (1, True)])
def test_conv_group_thin_input_equals_own_launches(dev, k, with_skip):
    """The grouped form of conv_thin.hip (the stem conv3x3 over the packed input planes, ncsnpp.py:183, and the input-skip conv1x1 + h of Combine,
    layerspp.py:44-59): three problems of one layer with their own batch sizes / image sizes / tensors in ONE launch - outputs and GroupNorm
    partials equal every problem's own storm_conv launch bit for bit."""
    from storm_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(80)
    Cout = 128
    w = ops.pack_conv_weight((torch.randn(Cout, 8, k, k, generator=g) * 0.3).to(dev), dtype)
    bias = torch.randn(Cout, generator=g).to(dev)
    problems = []
    for B, H, W in ((2, 16, 64), (1, 19, 45), (3, 8, 37)):
        x = torch.randn(B, 8, H, W, generator=g)
        x[:, 4:] = 0.0
        skip = nhwc(torch.randn(B, Cout, H, W, generator=g)).to(dtype).to(dev) if with_skip else None
        problems.append(([ops.Seg(nhwc(x).to(dtype).to(dev), w, k * k)], dict(bias=bias, skip=skip, scale=0.5 if with_skip else 1.0)))
    own = [ops.conv(segs, Cout, gn_partials=True, **kw) for segs, kw in problems]
    assert ops.conv_kernel_name(problems[0][0], Cout, **problems[0][1]).startswith("storm::conv_thin_kernel")
    outs, parts = ops.conv_group(problems, Cout, gn_partials=True)
    for p in range(3):
        assert torch.equal(outs[p], own[p][0]) and torch.equal(parts[p], own[p][1]), p


PIPE128_CASES = {
    # name: (B, H, W, Cout, (Ca, Cb) of the 3x3 operand, fused GroupNorm on it, (Sa, Sb) of the fused 1x1 shortcut or None, CUs)
    "plain": (2, 19, 45, 120, (72, 0), False, None, None),            # 3 chunks (the last ragged), ragged tile rows / columns
    "one_chunk": (1, 16, 32, 128, (32, 0), False, None, None),        # a single nine-tap chunk: the prefetch targets are terminators
    "one_chunk_tail": (1, 20, 40, 64, (24, 0), False, (40, 0), None), # one nine-tap chunk, then two one-tap chunks
    "block_tail": (2, 9, 35, 104, (104, 0), False, (72, 16), None),
    "gn_fused": (2, 10, 36, 40, (72, 56), True, None, None),
    "deep_k": (1, 18, 33, 120, (136, 88), True, (136, 72), None),     # 5 + 3 nine-tap chunks (fused GN), 5 + 3 one-tap chunks
    "plain@8": (2, 35, 70, 120, (72, 0), False, None, 8),             # 18 tiles on 8 persistent workgroups
    "block_tail@8": (2, 35, 70, 104, (40, 0), False, (72, 16), 8),
    "gn_fused@8": (2, 35, 70, 40, (72, 56), True, None, 8),
    "gn_fused:f16": (2, 10, 36, 40, (72, 56), True, None, None),
}


def test_split_k_small_call_rule(dev, switch):
    """conv_splitk_slices, clause (2) (round 5): a 3x3 layer with > 128 output channels splits its K loop when the WHOLE launch has at most
    64 unsplit workgroups - one utterance per call at the 32 x 64 (16 workgroups) and 64 x 128 (64) levels of NCSN++, the reference's own
    operating point (enhancement.py:66-72) - and not at the bench batch, where the same layers fill the chip; STORM_SPLITK_SMALL=0 leaves
    the per-image rule alone (rows then do not depend on the batch they ride in).  Result == the unsplit tile to fp32 summation order."""
    from storm_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    w = ops.pack_conv_weight((torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev), dtype)
    name = lambda B, H, W: ops.conv_kernel_name([ops.Seg(torch.zeros(B, H, W, 256, dtype=dtype, device=dev), w, 9)], 256)
    split, half, full = "storm::conv_pipe_splitk_kernel", "storm::conv_pipe_kernel<storm::bf16_t, 128, 8", "storm::conv_pipe_kernel<storm::bf16_t, 256, 8"
    assert name(1, 32, 64).startswith(split) and name(4, 32, 64).startswith(split) and name(1, 64, 128).startswith(split)
    assert name(5, 32, 64).startswith(half) and name(16, 32, 64).startswith(half) and name(2, 64, 128).startswith(half)
    assert name(1, 128, 256).startswith(half) and name(1, 256, 512).startswith(full)
    assert name(16, 8, 32).startswith(split)                       # (the image rule: 2 tiles per image, whatever the batch)
    switch("STORM_SPLITK_SMALL", 0)
    assert name(1, 32, 64).startswith(half) and name(1, 64, 128).startswith(half) and name(16, 8, 32).startswith(split)
    switch("STORM_SPLITK_SMALL", 1)
    x = torch.randn(1, 256, 32, 64, generator=g)
    seg = [ops.Seg(nhwc(x).to(dtype).to(dev), w, 9)]
    y, part = ops.conv(seg, 256, gn_partials=True)
    switch("STORM_SPLITK_SMALL", 0)
    y0, part0 = ops.conv(seg, 256, gn_partials=True)
    assert rel_l2(y.float().cpu(), y0.float().cpu()) < 3e-3 and not torch.equal(part, torch.zeros_like(part))
    assert torch.allclose(part.cpu(), part0.cpu(), rtol=2e-2, atol=2e-2 * float(part0.abs().max()))


def test_batch_invariant_mode_decides_per_image(dev, switch):
    """STORM_BATCH_INVARIANT=1 (the 256- / 128-cout tile choice looked at B x tiles, and the two tiles sum the fused GroupNorm
    partials in different orders): every launch decision that changes a summation order - choose_variant's ladder, the batch-ranged table,
    the small-call K split, the attention's key ranges - is then taken for ONE image whatever the batch.  Here: the kernel a layer gets is
    the same for every batch size at every level of NCSN++ (4-s and 10-s utterances), for both cout classes, with and without a fused
    shortcut, while the default rules do differ across those batch sizes; the network-level bit-equality is tests/test_net.py's."""
    from storm_amd import ops, _lib as L
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    packed = {}

    def name(B, H, W, cin, cout, sc):
        for key, shape in (((cout, cin, 3), (cout, cin, 3, 3)), ((cout, sc, 1), (cout, sc, 1, 1))):
            if key[1] and key not in packed:
                packed[key] = ops.pack_conv_weight((torch.randn(*shape, generator=g) * 0.03).to(dev), dtype)
        segs = [ops.Seg(torch.zeros(B, H, W, cin, dtype=dtype, device=dev), packed[(cout, cin, 3)], 9)]
        if sc:
            segs.append(ops.Seg(torch.zeros(B, H, W, sc, dtype=dtype, device=dev), packed[(cout, sc, 1)], 1))
        return ops.conv_kernel_name(segs, cout)
    layers = [(H, W, cin, cout, sc) for (H, W) in ((256, 512), (128, 256), (64, 128), (32, 64), (16, 32), (64, 320))
              for (cin, cout, sc) in ((256, 256, 0), (128, 128, 0), (256, 128, 256), (128, 256, 128))]
    batches = (1, 2, 3, 4, 8, 16)
    differs = sum(len({name(B, *l) for B in batches}) > 1 for l in layers)
    assert differs >= 4                                     # the default rules ARE decisions on the launch
    switch("STORM_BATCH_INVARIANT", 1)
    for l in layers:
        assert len({name(B, *l) for B in batches}) == 1, l
    assert L.lib().storm_attention_scratch_bytes(1, 2048, 256, L.BF16) == 0      # (no key-range split: one utterance would get one by default)
    switch("STORM_BATCH_INVARIANT", 0)
    assert L.lib().storm_attention_scratch_bytes(1, 2048, 256, L.BF16) > 0


@pytest.mark.parametrize("case", ["natural", "natural@8", "deep_k", "deep_k@8", "natural:f16", "one_slice_pair", "three_chunks", "tiny_images"])
def test_conv_split_k(dev, case, switch):
    """Split-K for 3x3 layers whose 128-cout tiles would leave most CUs idle (conv_pipe.hip: conv_pipe_splitk_kernel + splitk_combine_kernel;
    ncsnpp.py:460-470 - the deep levels of ncsnpplarge): K slices on separate workgroups write fp32 slabs, one combine pass sums them in
    slice order and applies bias / temb bias / skip / scale / rounding / GroupNorm partials.  "natural": the dispatcher's own choice
    (> 128 couts, few tiles) with ragged rows / columns / couts and a skip operand; "deep_k": 3 + 2 nine-tap chunks over a concat with a
    fused GroupNorm operand + 3 + 2 one-tap chunks of a fused shortcut (they ride with the last slice); "@8": eight resident workgroups
    walk all (tile, cout tile, slice) blocks; "three_chunks": uneven slices; "tiny_images": images smaller than a tile; against the
    unsplit tile (variant 9), F.conv2d, and itself (bit-reproducible)."""
    from storm_amd import ops
    cus = 0
    if case.endswith("@8"):
        cus, case = 8, case[:-2]
    dtype = torch.bfloat16
    if case.endswith(":f16"):
        dtype, case = torch.float16, case[:-4]
    g = torch.Generator().manual_seed(91)
    dd = lambda t: nhwc(t).to(dtype).to(dev)
    if case in ("natural", "one_slice_pair", "three_chunks", "tiny_images"):
        # three_chunks: 136 input channels = 3 chunks (the last one ragged) -> 2 slices of 1 and 2 chunks; tiny_images: 8 images of 4 x 16
        # pixels (half a tile high, half a tile wide: the deepest level of ncsnpplarge at configs[3])
        B, Cin, Cout, outC, H, W = {"natural": (2, 256, 280, 288, 9, 20), "one_slice_pair": (1, 128, 96, 96, 5, 20),
                                    "three_chunks": (2, 136, 160, 160, 9, 20), "tiny_images": (8, 256, 256, 256, 4, 16)}[case]
        x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        bias, tb = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
        sk = torch.randn(B, outC, H, W, generator=g)
        segs = [ops.Seg(dd(x), ops.pack_conv_weight(w.to(dev), dtype), 9)]
        kw = dict(bias=bias.to(dev), tbias=tb.to(dev), skip=dd(sk), scale=2 ** -0.5, outC=outC)
        ref = (F.conv2d(q(x, dtype), q(w, dtype), bias, padding=1) + tb[:, :, None, None] + q(sk, dtype)[:, :Cout]) * 2 ** -0.5
        tol_ref = 6e-3
    else:
        B, C0, Ca, Cb, Sa, Sb, Co, H, W = 1, 8, 136, 88, 136, 72, 264, 9, 20
        x0 = torch.randn(B, C0, H, W, generator=g)
        wa, wb = torch.randn(Ca, C0, 3, 3, generator=g) * 0.4, torch.randn(Cb, C0, 1, 1, generator=g) * 0.7
        w = torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.05
        sa, sb = torch.randn(B, Sa, H, W, generator=g), torch.randn(B, Sb, H, W, generator=g)
        w2 = torch.randn(Co, Sa + Sb, 1, 1, generator=g) * 0.1
        bias = torch.randn(Co, generator=g)
        gam, bet = 1 + 0.1 * torch.randn(Ca + Cb, generator=g), 0.1 * torch.randn(Ca + Cb, generator=g)
        xa, pa = ops.conv([ops.Seg(dd(x0), ops.pack_conv_weight(wa.to(dev), dtype), 9)], Ca, gn_partials=True)
        xb, pb = ops.conv([ops.Seg(dd(x0), ops.pack_conv_weight(wb.to(dev), dtype), 1)], Cb, gn_partials=True)
        _, ss = ops.gn_finalize(pa, pb, gamma=gam.to(dev), beta=bet.to(dev), count=H * W)
        segs = [ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, src_b=xb, gn_ss=ss, gn_silu=True),
                ops.Seg(dd(sa), ops.pack_conv_weight(w2.to(dev), dtype), 1, src_b=dd(sb))]
        kw = dict(bias=bias.to(dev), scale=0.5)
        xcat = torch.cat([nchw(xa.float().cpu()), nchw(xb.float().cpu())], 1)
        a = NR.silu(NR.group_norm(xcat, gam, bet))
        ref = (F.conv2d(q(a, dtype), q(w, dtype), padding=1) + F.conv2d(q(torch.cat([sa, sb], 1), dtype), q(w2, dtype))
               + bias[None, :, None, None]) * 0.5
        Cout, tol_ref = Co, 1e-2
    if cus:
        switch("STORM_CONV_CUS", cus)
    switch("STORM_CONV_VARIANT", 9)
    y9, part9 = ops.conv(segs, Cout, gn_partials=True, **kw)
    switch("STORM_CONV_VARIANT", 10 if case == "one_slice_pair" else -1)      # (96 couts on one cout tile: only a forced split applies)
    assert ops.conv_kernel_name(segs, Cout, **kw).startswith("storm::conv_pipe_splitk_kernel")
    y, part = ops.conv(segs, Cout, gn_partials=True, **kw)
    y2, part2 = ops.conv(segs, Cout, gn_partials=True, **kw)
    assert torch.equal(y, y2) and torch.equal(part, part2)
    yc = nchw(y.float().cpu())
    assert rel_l2(yc[:, :Cout], ref) < tol_ref
    if yc.shape[1] > Cout:                                  # padding channels: (0 + skip) * scale, as every conv kernel writes them
        assert torch.equal(y[..., Cout:], y9[..., Cout:])
    assert rel_l2(yc, nchw(y9.float().cpu())) < 2e-3        # (another fp32 summation order: a few values round the other way)
    assert part.shape == part9.shape
    assert torch.allclose(part.cpu(), part9.cpu(), rtol=2e-2, atol=2e-2 * float(part9.abs().max()))
    if y.shape[-1] % ops.gn_groups(y.shape[-1]) == 0:
        st, sref = ops.gn_finalize(part).cpu(), ops.gn_stats(y).cpu()
        assert torch.allclose(st, sref, rtol=2e-3, atol=2e-3 * float(sref.abs().max()))


def test_conv_dispatch_with_few_pixel_tiles(dev):
    """storm_conv's own choice (no switch) for a 3x3 layer with > 128 output channels: with fewer than 512 pixel tiles the pipelined
    kernel's 128-cout tile (16-bit operands; fp32: the 64-cout tile of conv_igemm.hip when 128-cout tiles would give <= 256
    workgroups), with many pixel tiles its 256-cout tile."""
    from storm_amd import ops
    w = ops.pack_conv_weight(torch.zeros(256, 64, 3, 3, device=dev), torch.bfloat16)
    small = [ops.Seg(torch.zeros(1, 32, 64, 64, dtype=torch.bfloat16, device=dev), w, 9)]
    assert ops.conv_kernel_name(small, 256) == "storm::conv_pipe_kernel<storm::bf16_t, 128, 8, 0>"
    large = [ops.Seg(torch.zeros(16, 64, 128, 64, dtype=torch.bfloat16, device=dev), w, 9)]
    assert ops.conv_kernel_name(large, 256) == "storm::conv_pipe_kernel<storm::bf16_t, 256, 8, 0>"
    w32 = ops.pack_conv_weight(torch.zeros(256, 64, 3, 3, device=dev), torch.float32)
    small32 = [ops.Seg(torch.zeros(1, 32, 64, 64, dtype=torch.float32, device=dev), w32, 9)]
    assert "1, 2, 2, false, true" in ops.conv_kernel_name(small32, 256)


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("case", list(PIPE128_CASES))
def test_conv_pipelined_128cout_kernel(dev, case, variant, switch):
    """The pipelined kernels for <= 128 output channels - conv_pipe128.hip (variant 4: 128 couts x 16 x 32 pixels, 8 waves, triple-
    buffered patches, 32-channel chunks) and conv_duo.hip (variant 5: 128 couts x 8 x 32 pixels, 4 waves, two workgroups per CU,
    staging and the fused GroupNorm transform inside the MFMA stream, one barrier per phase): plain 3x3, fused 1x1 shortcut over a concat, fused GroupNorm-apply operand, ragged
    sizes, 1 .. 8 nine-tap chunks, persistent tile walk, GroupNorm partials in the 8-row tile layout - on shapes the default
    dispatch would give to conv_igemm.hip."""
    from storm_amd import ops
    B, H, W, Co, (Ca, Cb), gn, one, cus = PIPE128_CASES[case]
    dtype = torch.float16 if case.endswith(":f16") else torch.bfloat16
    g = torch.Generator().manual_seed(131)
    dd = lambda t: t.to(dtype).to(dev)
    if gn:   # the operand is produced by convs (default dispatch) so that its GroupNorm partials exist
        x0 = torch.randn(B, 8, H, W, generator=g)
        wa, wb = torch.randn(Ca, 8, 3, 3, generator=g) * 0.4, torch.randn(max(Cb, 8), 8, 1, 1, generator=g) * 0.7
        gam, bet = 1 + 0.1 * torch.randn(Ca + Cb, generator=g), 0.1 * torch.randn(Ca + Cb, generator=g)
        xa, pa = ops.conv([ops.Seg(dd(nhwc(x0)), ops.pack_conv_weight(wa.to(dev), dtype), 9)], Ca, gn_partials=True)
        xb, pb = ops.conv([ops.Seg(dd(nhwc(x0)), ops.pack_conv_weight(wb.to(dev), dtype), 1)], Cb, gn_partials=True) if Cb else (None, None)
        _, ss = ops.gn_finalize(pa, pb, gamma=gam.to(dev), beta=bet.to(dev), count=H * W)
        xcat = torch.cat([nchw(t.float().cpu()) for t in (xa, xb) if t is not None], 1)
        a_ref = NR.silu(NR.group_norm(xcat, gam, bet))
    else:
        xcat = torch.randn(B, Ca + Cb, H, W, generator=g)
        xa, xb = dd(nhwc(xcat[:, :Ca])), (dd(nhwc(xcat[:, Ca:])) if Cb else None)
        ss, a_ref = None, xcat
    w = torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.05
    bias, tb = torch.randn(Co, generator=g), torch.randn(B, Co + 8, generator=g)
    segs = [ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, src_b=xb, gn_ss=ss, gn_silu=True)]
    ref = F.conv2d(q(a_ref, dtype), q(w, dtype), padding=1) + bias[None, :, None, None] + tb[:, 8:8 + Co, None, None]
    skip = None
    if one is not None:
        Sa, Sb = one
        s = torch.randn(B, Sa + Sb, H, W, generator=g)
        w2 = torch.randn(Co, Sa + Sb, 1, 1, generator=g) * 0.1
        segs.append(ops.Seg(dd(nhwc(s[:, :Sa])), ops.pack_conv_weight(w2.to(dev), dtype), 1, src_b=dd(nhwc(s[:, Sa:])) if Sb else None))
        ref = ref + F.conv2d(q(s, dtype), q(w2, dtype))
    else:
        skip = torch.randn(B, ops.round_up(Co, 8), H, W, generator=g)
        ref = ref + q(skip, dtype)[:, :Co]
        skip = dd(nhwc(skip))
    ref = ref * 0.5
    tbd = tb.to(dev)
    y_generic, part_generic = ops.conv(segs, Co, bias=bias.to(dev), tbias=tbd[:, 8:], skip=skip, scale=0.5, gn_partials=True)
    switch("STORM_CONV_VARIANT", variant)
    if cus:
        switch("STORM_CONV_CUS", cus)
    kname = ops.conv_kernel_name(segs, Co, bias=bias.to(dev), tbias=tbd[:, 8:], skip=skip, scale=0.5)
    if variant == 5 and not kname.startswith("storm::conv_duo_kernel"):
        pytest.skip("conv_duo.hip is not in the product library (LAB_NOTES 2.3): profiling library and the simulator only")
    assert kname.startswith({4: "storm::conv_pipe128_kernel", 5: "storm::conv_duo_kernel"}[variant])
    y, part = ops.conv(segs, Co, bias=bias.to(dev), tbias=tbd[:, 8:], skip=skip, scale=0.5, gn_partials=True)
    yc = nchw(y.float().cpu())
    assert rel_l2(yc[:, :Co], ref) < (1e-2 if gn else 6e-3)
    if yc.shape[1] > Co:
        assert float(yc[:, Co:].abs().max()) == 0.0
    # same result as the generic kernel up to the accumulation order; statistics partials in the same 8 x 32 tile layout
    assert rel_l2(yc, nchw(y_generic.float().cpu())) < 3e-3
    assert part.shape == part_generic.shape
    assert torch.allclose(part.cpu(), part_generic.cpu(), rtol=2e-2, atol=2e-2 * float(part_generic.abs().max()))
    st, sref = ops.gn_finalize(part).cpu(), ops.gn_stats(y).cpu()
    assert torch.allclose(st, sref, rtol=2e-3, atol=2e-3 * float(sref.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fused_block_tail(dev, dtype):
    """Conv_1 3x3 over h + Conv_2 1x1 over cat[xa, xb] + bias + temb bias, rescaled (layerspp.py:266-274)."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(2)
    B, H, W, Ca, Cb, Co = 2, 9, 35, 24, 16, 40
    h = torch.randn(B, Co, H, W, generator=g)
    xa, xb = torch.randn(B, Ca, H, W, generator=g), torch.randn(B, Cb, H, W, generator=g)
    w1 = torch.randn(Co, Co, 3, 3, generator=g) * 0.1
    w2 = torch.randn(Co, Ca + Cb, 1, 1, generator=g) * 0.2
    bias, tb = torch.randn(Co, generator=g), torch.randn(B, 64, generator=g)
    segs = [ops.Seg(nhwc(h).to(dtype).to(dev), ops.pack_conv_weight(w1.to(dev), dtype), 9),
            ops.Seg(nhwc(xa).to(dtype).to(dev), ops.pack_conv_weight(w2.to(dev), dtype), 1, src_b=nhwc(xb).to(dtype).to(dev))]
    tbd = tb.to(dev)
    y = ops.conv(segs, Co, bias=bias.to(dev), tbias=tbd[:, 8:], scale=1 / math.sqrt(2)).float().cpu()
    ref = (F.conv2d(q(h, dtype), q(w1, dtype), padding=1) + F.conv2d(q(torch.cat([xa, xb], 1), dtype), q(w2, dtype))
           + bias[None, :, None, None] + tb[:, 8:8 + Co, None, None]) / math.sqrt(2)
    assert rel_l2(nchw(y), ref) < tol(dtype, 2e-6, 6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_skip_and_batched_weights(dev, dtype):
    """identity skip epilogue, and per-batch 'weights' = an activation (attention q k^T), fp32 scores out."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(3)
    B, Lq, Cc = 2, 40, 24
    qq, kk = torch.randn(B, Lq, Cc, generator=g), torch.randn(B, Lq, Cc, generator=g)
    qd, kd = qq.to(dtype).to(dev).view(B, 1, Lq, Cc), kk.to(dtype).to(dev).view(B, 1, Lq, Cc)
    S = ops.conv([ops.Seg(qd, kd.view(B, Lq, Cc), 1, w_batched=True)], Lq, outC=Lq, scale=0.5, out_f32=True)
    assert S.dtype == torch.float32
    ref = torch.einsum("bic,bjc->bij", q(qq, dtype), q(kk, dtype)) * 0.5
    assert rel_l2(S.cpu().view(B, Lq, Lq), ref) < 2e-6
    x = torch.randn(B, 16, 6, 9, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) * 0.2
    xd = nhwc(x).to(dtype).to(dev)
    y = ops.conv([ops.Seg(xd, ops.pack_conv_weight(w.to(dev), dtype), 9)], 16, skip=xd, scale=2.0).float().cpu()
    ref = (F.conv2d(q(x, dtype), q(w, dtype), padding=1) + q(x, dtype)) * 2.0
    assert rel_l2(nchw(y), ref) < tol(dtype, 2e-6, 6e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,Cb", [(8, 0), (128, 0), (256, 128), (24, 16)])
def test_groupnorm_silu(dev, dtype, C, Cb):
    from storm_amd import ops
    g = torch.Generator().manual_seed(4)
    B, H, W = 2, 8, 16
    x = torch.randn(B, C + Cb, H, W, generator=g) * 1.5 + 0.3
    gam, bet = 1 + 0.1 * torch.randn(C + Cb, generator=g), 0.1 * torch.randn(C + Cb, generator=g)
    xa = nhwc(x[:, :C]).to(dtype).to(dev)
    xb = nhwc(x[:, C:]).to(dtype).to(dev) if Cb else None
    st = ops.gn_stats(xa, xb)
    y = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), xb=xb, silu=True).float().cpu()
    ref = NR.silu(NR.group_norm(q(x, dtype), gam, bet))
    assert rel_l2(nchw(y), ref) < tol(dtype, 2e-6, 5e-3)
    y2 = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), xb=xb, silu=False).float().cpu()
    assert rel_l2(nchw(y2), NR.group_norm(q(x, dtype), gam, bet)) < tol(dtype, 2e-6, 5e-3)


def test_groupnorm_golden(dev, golden):
    """the reference's own GroupNorm+SiLU outputs (tests/golden/f1_ops.npz)"""
    from storm_amd import ops
    g = golden["f1_ops"]
    for C in (8, 128, 384):
        x = torch.from_numpy(g[f"gn{C}_x"])
        xa = nhwc(x).to(dev)
        st = ops.gn_stats(xa)
        y = ops.gn_apply(xa, st, torch.from_numpy(g[f"gn{C}_w"]).to(dev), torch.from_numpy(g[f"gn{C}_b"]).to(dev))
        assert rel_l2(nchw(y.cpu()), g[f"gn{C}_y"]) < 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("resample", [1, 2])
@pytest.mark.parametrize("shape", [(2, 24, 6, 10), (2, 72, 20, 36), (1, 160, 38, 70), (1, 128, 6, 20), (2, 256, 10, 12)])
def test_groupnorm_fir_fused(dev, dtype, resample, shape, switch):
    """h = FIR(SiLU(GN(x))) and x = FIR(x) of a BigGAN up/down block in one pass (layerspp.py:243-255); second / third shape: two
    and three channel groups (the last ragged), several row strips and column blocks per image - every thread walks down a strip
    with the rows it shares with the previous output row carried in registers; the last two: the channel counts of the networks, which
    run with 16 / 32 slots of a pixel per workgroup (same bits as the 8-slot layout, STORM_GN_WIDE = 0)."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, H, W = shape
    x = torch.randn(B, C, H, W, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xa = nhwc(x).to(dtype).to(dev)
    st = ops.gn_stats(xa)
    act, raw = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), resample=resample)
    fir = NR.fir_up2 if resample == 1 else NR.fir_down2
    assert rel_l2(nchw(act.float().cpu()), fir(NR.silu(NR.group_norm(q(x, dtype), gam, bet)))) < tol(dtype, 2e-6, 5e-3)
    assert rel_l2(nchw(raw.float().cpu()), fir(q(x, dtype))) < tol(dtype, 2e-6, 5e-3)
    switch("STORM_GN_WIDE", 0)
    act8, raw8 = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), resample=resample)
    assert torch.equal(act, act8) and torch.equal(raw, raw8)
    for rows in (4, 8, 16):          # rows per strip (strip_rows picks 4 ... 16 by the size of the call): the same bits whatever the strips
        switch("STORM_GN_ROWS", rows)
        actr, rawr = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), resample=resample)
        assert torch.equal(act, actr) and torch.equal(raw, rawr)
    switch("STORM_GN_ROWS", 0)
    if resample == 2:                # the down-sampling kernel with the activation shared between neighbouring threads through LDS (full launches) and
        for share in (1, 2):         # the barrier-free one (small calls): the same bits, whatever the strips
            switch("STORM_GN_DOWN_SHARE", share)
            for rows in (0, 4):
                switch("STORM_GN_ROWS", rows)
                acts, raws = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), resample=resample)
                assert torch.equal(act, acts) and torch.equal(raw, raws), (share, rows)
            switch("STORM_GN_WIDE", 1)
            acts, raws = ops.gn_apply(xa, st, gam.to(dev), bet.to(dev), resample=resample)
            assert torch.equal(act, acts) and torch.equal(raw, raws), (share, "wide")
            switch("STORM_GN_WIDE", 0)


def test_fir_golden(dev, golden):
    from storm_amd import ops
    g = golden["f1_ops"]
    x = torch.from_numpy(g["fir_x"])                       # [2,5,8,12]: pad channels to 8
    xp = torch.cat([x, torch.zeros(2, 3, 8, 12)], 1)
    up = ops.fir_up2(nhwc(xp).to(dev)).cpu()
    dn = ops.fir_down2(nhwc(xp).to(dev)).cpu()
    assert rel_l2(nchw(up)[:, :5], g["fir_up"]) < 1e-6
    assert rel_l2(nchw(dn)[:, :5], g["fir_down"]) < 1e-6
    add = torch.randn(2, 8, 16, 24)
    up2 = ops.fir_up2(nhwc(xp).to(dev), add=nhwc(add).to(dev)).cpu()
    assert rel_l2(nchw(up2)[:, :5], torch.from_numpy(g["fir_up"]) + add[:, :5]) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_pyramids_equal_their_chains(dev, dtype):
    """csrc/pyramid.hip: the progressive input pyramid (pack + every FIR x2 down step) and the output pyramid (every FIR x2 up step + `+ ph` + the
    head) as ONE launch each, against the chains of launches they replace - storm_pack_input / storm_fir_down2 and storm_fir_up2 (add) /
    storm_output_head, themselves pinned by the reference's fixtures (test_fir_golden, the network fixtures) - BIT for bit: every level goes through
    the same taps in the same order and is rounded to the storage type where the chain stored it.  Shapes with several workgroup tiles per axis,
    one to three steps per launch, the continuation launch of a pyramid deeper than three steps (ncsnpplarge: six), seven output levels."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(77)
    big = dev.type != "cpu"
    for (B, F_, T_, n_levels) in ([(2, 64, 128, 4), (1, 32, 192, 3), (2, 32, 64, 2), (1, 8, 24, 1)] if big else [(1, 64, 128, 4), (1, 16, 64, 2)]):
        cplx = [torch.complex(torch.randn(B, F_, T_, generator=g), torch.randn(B, F_, T_, generator=g)).to(dev) for _ in range(2)]
        chain = [ops.pack_input(cplx, dtype)]
        for _ in range(n_levels - 1):
            chain.append(ops.fir_down2(chain[-1]))
        got = ops.input_pyramid(cplx, dtype, n_levels)
        assert len(got) == n_levels
        for k in range(n_levels):
            assert got[k].shape == chain[k].shape and torch.equal(got[k], chain[k]), (B, F_, T_, n_levels, k)
        if n_levels >= 3:                                   # a pyramid continued from a given level (levels 3 .. 6 of ncsnpplarge)
            cont = ops.input_pyramid(None, dtype, n_levels - 1, level0=chain[1].clone())
            for k in range(1, n_levels):
                assert torch.equal(cont[k - 1], chain[k]), ("continued", k)
    t = (0.2 + 0.7 * torch.rand(2, generator=g)).to(dev)
    for (B, F_, T_, n_levels, cin) in ([(2, 64, 192, 4, 4), (1, 64, 128, 7, 6), (2, 40, 72, 1, 2), (1, 96, 64, 3, 4)] if big else [(1, 64, 128, 4, 4), (1, 64, 64, 7, 2)]):
        phs = []
        for k in range(n_levels):
            ph = torch.zeros(B, F_ >> k, T_ >> k, 8)
            ph[..., :cin] = torch.randn(B, F_ >> k, T_ >> k, cin, generator=g)
            phs.append(ph.to(dtype).to(dev))
        W, bias = (0.5 * torch.randn(2, cin, generator=g)).to(dev), torch.randn(2, generator=g).to(dev)
        p = phs[-1]
        for k in range(n_levels - 2, -1, -1):
            p = ops.fir_up2(p, add=phs[k])
        for tt, neg in ((t[:B].contiguous(), True), (None, False)):
            want = ops.output_head(p, tt, W, bias, neg)
            got = ops.output_pyramid(phs, tt, W, bias, neg)
            assert torch.equal(torch.view_as_real(got), torch.view_as_real(want)), (B, F_, T_, n_levels, neg)
    with pytest.raises(Exception):
        ops.input_pyramid([torch.zeros(1, 12, 16, dtype=torch.complex64, device=dev)], dtype, 4)       # 12 rows: not divisible by 2^3


@pytest.mark.parametrize("name", ["up2", "down2", "mixed", "updown"])
def test_upfirdn2d_reference_argument_list(dev, golden, name):
    """storm_upfirdn2d = the reference's one native-op ABI with its own argument list (op/upfirdn2d.cpp:12-22: input [N,H,W,1], kernel
    [kh,kw], up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1) against what upfirdn2d_native returned (fixture F17): the
    two parameter sets of the network and two that no layer uses; 16-bit planes against the fp32 result within their rounding."""
    from storm_amd import ops
    g = golden["f17_upfirdn2d"]
    x, k, args = torch.from_numpy(g[f"{name}_x"]), torch.from_numpy(g[f"{name}_k"]), [int(v) for v in g[f"{name}_args"]]
    want = torch.from_numpy(g[f"{name}_y"])
    y = ops.upfirdn2d(x[..., None].contiguous().to(dev), k.to(dev), *args).cpu()
    assert y.shape == want.shape + (1,) and rel_l2(y[..., 0], want) < 1e-6
    for dt in (torch.bfloat16, torch.float16):
        y16 = ops.upfirdn2d(x[..., None].contiguous().to(dt).to(dev), k.to(dev), *args).float().cpu()
        assert rel_l2(y16[..., 0], want) < 8e-3
    with pytest.raises(Exception):
        ops.upfirdn2d(x[..., None].contiguous().to(dev), k.to(dev), 0, 1, 1, 1, 0, 0, 0, 0)        # up_x = 0
    with pytest.raises(ValueError):
        ops.upfirdn2d(x.to(dev), k.to(dev))                                                        # not [N, H, W, 1]


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_rows(dev, dtype):
    from storm_amd import ops
    g = torch.Generator().manual_seed(6)
    s = torch.randn(3, 5, 200, generator=g) * 4
    s[0, 0, 7] = 60.0                                         # a spike: stable max subtraction
    p = ops.softmax_rows(s.to(dev), dtype).float().cpu()
    assert rel_l2(p, F.softmax(s, -1)) < tol(dtype, 1e-6, 4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_input_and_output_head(dev, dtype):
    from storm_amd import ops
    g = torch.Generator().manual_seed(7)
    B, Fq, T = 2, 8, 16
    zs = [torch.randn(B, Fq, T, dtype=torch.complex64, generator=g) for _ in range(3)]
    x = ops.pack_input([z.to(dev) for z in zs], dtype).float().cpu()
    ref = 2 * torch.stack([c for z in zs for c in (z.real, z.imag)], -1) - 1
    assert rel_l2(x[..., :6], q(ref, dtype)) < 1e-7 and float(x[..., 6:].abs().max()) == 0
    pyr = torch.randn(B, Fq, T, 8, generator=g)
    Wt, b, t = torch.randn(2, 6, 1, 1, generator=g), torch.randn(2, generator=g), torch.tensor([0.8, 0.05])
    out = ops.output_head(pyr.to(dtype).to(dev), t.to(dev), Wt.to(dev), b.to(dev), negate=True).cpu()
    h = q(pyr, dtype)[..., :6] / t[:, None, None, None]
    r = -(torch.einsum("bftc,oc->bfto", h, Wt.view(2, 6)) + b)
    assert rel_l2(torch.view_as_real(out), r) < 2e-6
    out2 = ops.output_head(pyr.to(dtype).to(dev), None, Wt.to(dev), b.to(dev), negate=False).cpu()
    r2 = torch.einsum("bftc,oc->bfto", q(pyr, dtype)[..., :6], Wt.view(2, 6)) + b
    assert rel_l2(torch.view_as_real(out2), r2) < 2e-6


def test_time_embedding_and_dense(dev):
    from storm_amd import ops
    g = torch.Generator().manual_seed(8)
    nf = 16
    sd = {"0.W": torch.randn(nf, generator=g) * 16, "1.weight": torch.randn(4 * nf, 2 * nf, generator=g) * 0.2,
          "1.bias": torch.randn(4 * nf, generator=g) * 0.1, "2.weight": torch.randn(4 * nf, 4 * nf, generator=g) * 0.2,
          "2.bias": torch.randn(4 * nf, generator=g) * 0.1}
    t = torch.tensor([1.0, 0.5, 0.03])
    d = {k: v.to(dev) for k, v in sd.items()}
    at = ops.time_embedding(t.to(dev), d["0.W"], d["1.weight"], d["1.bias"], d["2.weight"], d["2.bias"])
    ref = NR.silu(NR.time_embedding(NR._SD(sd), None, t))
    assert rel_l2(at.cpu(), ref) < 2e-5
    Wd, bd = torch.randn(37, 4 * nf, generator=g) * 0.2, torch.randn(37, generator=g)
    o = ops.dense(at, Wd.to(dev), bd.to(dev)).cpu()
    assert rel_l2(o, F.linear(at.cpu(), Wd, bd)) < 2e-6
    # rows wider than one register tile of the kernel (nf = 192 / 256 -> K = 4 nf = 768 / 1024; --nf is a reference flag), ragged K
    for K in (768, 1024, 520):
        xw, Ww, bw = torch.randn(11, K, generator=g), torch.randn(21, K, generator=g) * 0.05, torch.randn(21, generator=g)
        ow = ops.dense(xw.to(dev), Ww.to(dev), bw.to(dev)).cpu()
        assert rel_l2(ow, F.linear(xw, Ww, bw)) < 2e-6


# ---------------------------------------------------------------- SDE steps ---------------
def _sde():
    return SR.OUVE(1.5, 0.05, 0.5, N=30)


def test_sde_steps_vs_oracle(dev):
    from storm_amd import ops
    g = torch.Generator().manual_seed(9)
    sh = (3, 1, 8, 16)
    x, y, s, z = [torch.randn(sh, dtype=torch.complex64, generator=g) * 0.5 for _ in range(4)]
    t = torch.tensor([1.0, 0.5, 0.03])
    sde = _sde()
    score = lambda *_: s
    # prior
    xp = ops.ouve_prior(sde, y.to(dev), z=z.to(dev)).cpu()
    assert rel_l2(xp, sde.prior(y, z)) < 2e-7
    # ALD corrector
    xa, xm = ops.ouve_ald_step(sde, x.clone().to(dev), s.to(dev), t.to(dev), 0.5, z=z.to(dev))
    r, rm = SR.ald_step(sde, score, x, t, y, z, 0.5)
    assert rel_l2(xa.cpu(), r) < 3e-7 and rel_l2(xm.cpu(), rm) < 3e-7
    # reverse diffusion / euler-maruyama predictors
    for kind, fn in ((0, SR.revdiff_step), (1, SR.euler_maruyama_step)):
        xa, xm = ops.ouve_predictor_step(sde, x.clone().to(dev), s.to(dev), y.to(dev), t.to(dev), kind=kind, z=z.to(dev))
        r, rm = fn(sde, score, x, t, y, z)
        assert rel_l2(xa.cpu(), r) < 3e-7 and rel_l2(xm.cpu(), rm) < 3e-7
    xa, xm = ops.ouve_predictor_step(sde, x.clone().to(dev), s.to(dev), y.to(dev), t.to(dev), kind=0, noise_free=True)
    assert torch.equal(xa.cpu(), xm.cpu()) and rel_l2(xm.cpu(), SR.revdiff_step(sde, score, x, t, y, z)[1]) < 3e-7
    # langevin (batch-mean norms)
    xa, xm = ops.langevin_step(x.clone().to(dev), s.to(dev), z.to(dev), 0.5)
    r, rm = SR.langevin_step(sde, score, x, t, y, z, 0.5)
    assert rel_l2(xa.cpu(), r) < 3e-6 and rel_l2(xm.cpu(), rm) < 3e-6


def test_complex_randn_statistics(dev):
    from storm_amd import ops
    n = 1 << 16
    z = ops.complex_randn((n,), dev, seed=1234, offset=0).cpu()
    z2 = ops.complex_randn((n,), dev, seed=1234, offset=0).cpu()
    z3 = ops.complex_randn((n,), dev, seed=1234, offset=1).cpu()
    assert torch.equal(z, z2) and not torch.equal(z, z3)            # counter based: reproducible
    for comp in (z.real, z.imag):                                    # N(0, 1/2) per component
        assert abs(float(comp.mean())) < 0.01 and abs(float(comp.var()) - 0.5) < 0.01
    assert abs(float((z.real * z.imag).mean())) < 0.01
    kurt = float((z.real ** 4).mean() / (z.real ** 2).mean() ** 2)
    assert abs(kurt - 3.0) < 0.1
    # in-kernel noise path of a step == explicit noise with the same (seed, offset)
    sde = _sde()
    y = torch.zeros(2, 1, 8, 16, dtype=torch.complex64)
    zz = ops.complex_randn(y.shape, dev, seed=5, offset=3)
    a = ops.ouve_prior(sde, y.to(dev), z=None, seed=5, offset=3).cpu()
    b = ops.ouve_prior(sde, y.to(dev), z=zz).cpu()
    assert torch.equal(a, b)


# ---------------------------------------------------------------- spectral ----------------
@pytest.mark.parametrize("fac", [0.15, 0.33])
def test_stft_istft_golden(dev, golden, fac):
    from storm_amd import ops
    g = golden["f5_frontend"]
    key = f"L8000_f{int(fac * 100)}"
    y = torch.from_numpy(g[f"{key}_y"])                                # [1, 8000]
    yd = y.to(dev)
    peak = ops.peak_abs(yd)
    assert float(peak.cpu()) == float(y.abs().max())
    Y = ops.stft(yd, peak, spec_factor=fac, spec_abs_exponent=0.5, pad_to=64)
    assert Y.shape == (1, 256, 64)
    assert rel_l2(Y.cpu(), torch.from_numpy(g[f"{key}_Y"])[0]) < 5e-6
    w = ops.istft(torch.from_numpy(g[f"{key}_Y"])[0].to(dev), 8000, None, spec_factor=fac, spec_abs_exponent=0.5)
    assert rel_l2(w.cpu(), g[f"{key}_wav"]) < 5e-6


@pytest.mark.parametrize("tag", ["sqrthann", "lin", "sq667", "n254", "hop256"])
def test_data_module_settings_vs_reference_golden(dev, golden, tag):
    """F20: the data module's settings beside the defaults (data_module.py:19-25, 142-148, 182-223; the reference CLI's --window /
    --n_fft / --hop_length / --spec_factor / --spec_abs_exponent): sqrt-Hann window, exponent 1, another exponent and factor, a 254-point
    transform, a 256-sample hop - SpecsDataModule.stft / spec_fwd / spec_back / istft against the reference's on the same signal."""
    from oracle.make_golden import F20_CASES
    from storm_amd.data_module import SpecsDataModule
    g = golden["f20_data_module"]
    dm = SpecsDataModule(gpu=False, **F20_CASES[tag])
    y = torch.from_numpy(g["y"]).to(dev)
    Y = dm.spec_fwd(dm.stft(y))
    want = torch.from_numpy(g[f"{tag}_Y"])
    assert Y.shape == want.shape
    e1 = rel_l2(Y.cpu(), want)
    w = dm.istft(dm.spec_back(want.to(dev)), 3000)
    e2 = rel_l2(w.cpu(), g[f"{tag}_wav"])
    print(f"data module {tag}: spec rel-L2 vs reference {e1:.2e}, wav {e2:.2e}")
    assert e1 < 5e-6 and e2 < 5e-6


def test_stft_batched_ragged_vs_oracle(dev):
    """batched front end == per-utterance reference calls; odd length, exponent 1 path"""
    from storm_amd import ops
    g = torch.Generator().manual_seed(11)
    y = torch.randn(3, 5003, generator=g) * 0.1
    Y = ops.stft(y.to(dev), None, spec_factor=1.0, spec_abs_exponent=1.0).cpu()
    ref = FR.stft(y)
    assert Y.shape == ref.shape and rel_l2(Y, ref) < 5e-6
    peak = ops.peak_abs(y.to(dev))
    Y2 = ops.stft(y.to(dev), peak, spec_factor=0.15, spec_abs_exponent=0.5, pad_to=64)
    for b in range(3):
        Yb, nf, T0 = FR.wav_to_spec(y[b:b + 1])
        assert rel_l2(Y2[b].cpu(), Yb[0, 0]) < 5e-6
    w = ops.istft(Y2, 5003, peak, spec_factor=0.15, spec_abs_exponent=0.5).cpu()
    for b in range(3):
        Yb, nf, T0 = FR.wav_to_spec(y[b:b + 1])
        assert rel_l2(w[b], FR.spec_to_wav(Yb, nf, T0)) < 5e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_fused_gn_statistics(dev, dtype, k):
    """per-tile (sum, sumsq) partials from the conv epilogue + finalize == a statistics pass over the output,
    also for the channel concat of two producers (up-path skip connections)."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(12)
    B, Cin, Cout, H, W = 2, 16, 40, 9, 37
    x = torch.randn(B, Cin, H, W, generator=g)
    w1, w2 = torch.randn(Cout, Cin, k, k, generator=g) * 0.3, torch.randn(24, Cin, k, k, generator=g) * 0.3
    xd = nhwc(x).to(dtype).to(dev)
    y1, p1 = ops.conv([ops.Seg(xd, ops.pack_conv_weight(w1.to(dev), dtype), k * k)], Cout, gn_partials=True)
    y2, p2 = ops.conv([ops.Seg(xd, ops.pack_conv_weight(w2.to(dev), dtype), k * k)], 24, gn_partials=True, scale=0.5)
    rtol = 1e-5 if dtype == torch.float32 else 2e-3
    st = ops.gn_finalize(p1).cpu()
    ref = ops.gn_stats(y1).cpu()
    assert torch.allclose(st, ref, rtol=rtol, atol=rtol * float(ref.abs().max()))
    st = ops.gn_finalize(p1, p2).cpu()
    ref = ops.gn_stats(y1, y2).cpu()
    assert torch.allclose(st, ref, rtol=rtol, atol=rtol * float(ref.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fused_groupnorm_apply(dev, dtype):
    """conv3x3(SiLU(GN(cat[xa, xb]))) with the normalisation applied inside the conv's operand load
    (statistics from the producers' epilogues) == GroupNorm -> SiLU -> conv of the oracle."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(13)
    B, C0, Ca, Cb, Co, H, W = 2, 8, 24, 16, 40, 10, 36
    x0 = torch.randn(B, C0, H, W, generator=g)
    wa, wb = torch.randn(Ca, C0, 3, 3, generator=g) * 0.4, torch.randn(Cb, C0, 1, 1, generator=g) * 0.7
    w = torch.randn(Co, Ca + Cb, 3, 3, generator=g) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(Ca + Cb, generator=g), 0.1 * torch.randn(Ca + Cb, generator=g)
    x0d = nhwc(x0).to(dtype).to(dev)
    xa, pa = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wa.to(dev), dtype), 9)], Ca, gn_partials=True)
    xb, pb = ops.conv([ops.Seg(x0d, ops.pack_conv_weight(wb.to(dev), dtype), 1)], Cb, gn_partials=True)
    st, ss = ops.gn_finalize(pa, pb, gamma=gam.to(dev), beta=bet.to(dev), count=H * W)
    y = ops.conv([ops.Seg(xa, ops.pack_conv_weight(w.to(dev), dtype), 9, src_b=xb, gn_ss=ss, gn_silu=True)], Co)
    xcat = torch.cat([nchw(xa.float().cpu()), nchw(xb.float().cpu())], 1)
    a = NR.silu(NR.group_norm(xcat, gam, bet))
    ref = F.conv2d(q(a, dtype), q(w, dtype), padding=1)
    assert rel_l2(nchw(y.float().cpu()), ref) < tol(dtype, 5e-6, 1e-2)


def test_conv_wait_placement_under_late_dma_landing():
    """The LDS-DMA kernels again, with the simulator's DMA queue in its 'late' mode: a copy becomes visible only when
    the issuing lane's counted vm_wait retires it, so a fragment read placed before the wait + barrier that publishes
    its data reads stale LDS and fails parity.  (The default mode lands copies at once, which exposes the opposite,
    write-after-read, class of hazards.)  Runs in a subprocess: the mode is fixed per loaded simulator library."""
    import os
    import subprocess
    import sys
    if os.environ.get("STORM_SIM_DMA") == "late":
        pytest.skip("already inside the late-landing run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the inner run is this suite's longest item: two workers of its own - tests/conftest.py - beside the outer run's)
    env = dict(os.environ, STORM_SIM_DMA="late", PYTHONPATH=root, STORM_TEST_WORKERS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_ops.py"), "-q", "-x", "-m", "not gpu",
                        "-k", "conv and not late_dma", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_attention_group_equals_own_unsplit_calls(dev, dt, switch):
    """storm_attention_group: the fused attention (AttnBlockpp, layerspp.py:82-86) of three problems with their own batch sizes and sequence
    lengths (ragged query / key tiles) in ONE launch - every problem equals its own UNSPLIT storm_attention call bit for bit (the grouped
    launch never splits the key loop; a small call on its own would, and then agrees to fp32 rounding of the merge)."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(44)
    Cc = 64
    bias = (0.1 * torch.randn(Cc, generator=g)).to(dev)
    problems = []
    for B, Lq in ((2, 100), (1, 260), (3, 33)):
        q, k, v = (torch.randn(B, Lq, Cc, generator=g) for _ in range(3))
        ldv = ops.round_up(Lq, 8)
        vT = torch.zeros(B, Cc, ldv)
        vT[:, :, :Lq] = v.transpose(1, 2)
        problems.append(((q * 1.5).to(dt).to(dev), k.to(dt).to(dev), vT.to(dt).to(dev)))
    switch("STORM_ATTN_SPLIT", 1)
    own = [ops.attention(q, k, vT, bias, Cc ** -0.5) for q, k, vT in problems]
    outs = ops.attention_group(problems, bias, Cc ** -0.5)
    for p in range(3):
        assert torch.equal(outs[p], own[p]), p
    switch("STORM_ATTN_SPLIT", 0)
    split = [ops.attention(q, k, vT, bias, Cc ** -0.5) for q, k, vT in problems]          # the small-call rule may split these
    for p in range(3):
        assert rel_l2(outs[p].float().cpu(), split[p].float().cpu()) < (8e-3 if dt == torch.bfloat16 else 1.5e-3)
    with pytest.raises(Exception):
        ops.attention_group([tuple(t.float() for t in problems[0])], bias, 1.0)       # fp32: outside the grouped kernel


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 8e-3), (torch.float16, 1.5e-3), (torch.float32, 2e-6)])
@pytest.mark.parametrize("C,Lq", [(32, 100), (64, 32), (256, 70)])
def test_fused_attention(dev, C, Lq, dt, tol):
    """storm_attention (flash style, online softmax) == softmax(q k^T / sqrt C) v + b_v of AttnBlockpp (layerspp.py:82-86),
    ragged key / query tiles, zero-padded v^T rows; bf16 / fp16 operands (P and the output are rounded to the 16-bit type)
    and the fp32 parity path (exact-fp32 MFMA)."""
    from storm_amd import ops
    g = torch.Generator().manual_seed(40 + C)
    B = 2
    q, k, v = (torch.randn(B, Lq, C, generator=g) for _ in range(3))
    q = q * 1.5                                             # sharper distributions: the running max must move between tiles
    bias = 0.1 * torch.randn(C, generator=g)
    ldv = ops.round_up(Lq, 8)
    vT = torch.zeros(B, C, ldv)
    vT[:, :, :Lq] = v.transpose(1, 2)
    rd = lambda t: t.to(dt)
    out = ops.attention(rd(q).to(dev), rd(k).to(dev), rd(vT).to(dev), bias.to(dev), C ** -0.5).float().cpu()
    w = torch.softmax(torch.einsum("bic,bjc->bij", rd(q).double(), rd(k).double()) * C ** -0.5, -1)
    ref = (torch.einsum("bij,bjc->bic", w, rd(v).double()) + bias).float()
    assert rel_l2(out, ref) < tol


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 8e-3), (torch.float16, 1.5e-3)])
@pytest.mark.parametrize("C,Lq,S", [(32, 300, 2), (64, 260, 8), (256, 200, 4), (256, 70, 2)])
def test_fused_attention_key_split(dev, C, Lq, S, dt, tol, switch):
    """attention.hip, round 5: a call whose query blocks leave most CUs idle (one utterance: 16 workgroups, each a chain of 64 key tiles)
    splits the KEY loop into S ranges on S x the workgroups (partial accumulator + running maximum / sum per range in caller scratch) and
    merges them in a second launch - the softmax of layerspp.py:82-86 over all keys, exactly.  Forced here at small L (ragged last
    tiles, uneven ranges, more ranges than whole tiles allow -> unsplit): == the reference formula, == the unsplit kernel to the
    rounding of the 16-bit output, bit-reproducible; and the rule's own choice for one / two / sixteen utterances at L = 2048."""
    from storm_amd import ops
    from storm_amd import _lib as L
    g = torch.Generator().manual_seed(60 + C)
    B = 2
    q, k, v = (torch.randn(B, Lq, C, generator=g) for _ in range(3))
    q = q * 1.5
    bias = 0.1 * torch.randn(C, generator=g)
    ldv = ops.round_up(Lq, 8)
    vT = torch.zeros(B, C, ldv)
    vT[:, :, :Lq] = v.transpose(1, 2)
    rd = lambda t: t.to(dt)
    args = (rd(q).to(dev), rd(k).to(dev), rd(vT).to(dev), bias.to(dev), C ** -0.5)
    switch("STORM_ATTN_SPLIT", 1)
    whole = ops.attention(*args).float().cpu()
    switch("STORM_ATTN_SPLIT", S)
    ntiles = -(-Lq // 32)
    assert (L.lib().storm_attention_scratch_bytes(B, Lq, C, L.dt(dt)) > 0) == (S <= ntiles)
    out = ops.attention(*args)
    assert torch.equal(out, ops.attention(*args))
    out = out.float().cpu()
    w = torch.softmax(torch.einsum("bic,bjc->bij", rd(q).double(), rd(k).double()) * C ** -0.5, -1)
    ref = (torch.einsum("bij,bjc->bic", w, rd(v).double()) + bias).float()
    assert rel_l2(out, ref) < tol and rel_l2(out, whole) < tol
    switch("STORM_ATTN_SPLIT", 0)
    nb = lambda b: L.lib().storm_attention_scratch_bytes(b, 2048, 256, L.dt(dt))
    assert nb(1) == 8 * 1 * 2048 * 258 * 4 and nb(2) == 8 * 2 * 2048 * 258 * 4 and nb(4) == 4 * 4 * 2048 * 258 * 4 and nb(8) == 0 and nb(16) == 0
    assert L.lib().storm_attention_scratch_bytes(1, 2048, 256, L.F32) == 0


@pytest.mark.gpu
def test_attention_block_L2048_vs_reference_golden(golden):
    """AttnBlockpp at the bench shape (256 channels, 32 x 64 = 2048 positions; layerspp.py:60-91) against the REFERENCE's
    output (fixture F8), composed from the C-ABI ops exactly as the planner does: GroupNorm -> NIN q, k (1x1 GEMMs), v^T by
    the swapped GEMM -> storm_attention -> NIN_3 + skip, rescaled.  bf16 operands."""
    import hashlib
    from storm_amd import ops
    from tests.backend import setup_backend
    from tests.test_net import seeded_input
    dev = setup_backend("hip")
    g = golden["f8_bench_shape"]
    x = seeded_input((1, 256, 32, 64), 810, 1.0, torch.float32)
    assert hashlib.sha256(x.contiguous().numpy().tobytes()).hexdigest() == str(g["attn_xhash"])
    C, Lp, dt = 256, 2048, torch.bfloat16
    P = {k[len("attn_"):]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith("attn_") and k not in ("attn_y", "attn_xhash")}
    xb = nhwc(x).to(dt).to(dev).repeat(2, 1, 1, 1)           # batch 2: rows must agree
    st = ops.gn_stats(xb)
    h = ops.gn_apply(xb, st, P["GroupNorm_0.weight"], P["GroupNorm_0.bias"], silu=False)
    hl = h.reshape(2, 1, Lp, C)
    W = [ops.pack_matrix(P[f"NIN_{i}.W"], dt, transpose=True) for i in range(4)]
    q = ops.conv([ops.Seg(hl, W[0], 1)], C, bias=P["NIN_0.b"])
    k = ops.conv([ops.Seg(hl, W[1], 1)], C, bias=P["NIN_1.b"])
    vT = ops.conv([ops.Seg(W[2].reshape(1, 1, C, C), hl.reshape(2, Lp, C), 1, w_batched=True, src_bstride=0)], Lp, outC=Lp, B=2, H=1, W=C)
    o = ops.attention(q.reshape(2, Lp, C), k.reshape(2, Lp, C), vT.reshape(2, C, Lp), P["NIN_2.b"], C ** -0.5)
    y = ops.conv([ops.Seg(o.reshape(2, 1, Lp, C), W[3], 1)], C, bias=P["NIN_3.b"], skip=xb.reshape(2, 1, Lp, C), scale=2 ** -0.5)
    y = nchw(y.reshape(2, 32, 64, C).float().cpu())
    err = rel_l2(y[:1], g["attn_y"])
    print(f"AttnBlockpp L=2048 bf16 (fused attention) vs reference: rel-L2 {err:.3e}")
    assert err < 2e-2 and torch.equal(y[0], y[1])


@pytest.mark.parametrize("tag", ["plain", "up", "down", "widen"])
def test_resblock_vs_reference_golden(dev, golden, tag):
    """ResnetBlockBigGANpp (layerspp.py:222-274: plain / up / down / channel-changing) against the REFERENCE's outputs
    (fixture F1 res_*), composed from the C-ABI ops the way the planner does: GroupNorm(+SiLU)(+FIR of h and x in one
    pass) -> Conv_0 + bias + Dense_0(act(temb)) with GroupNorm statistics from its epilogue -> GroupNorm+SiLU ->
    Conv_1 (+ fused 1x1 Conv_2 shortcut | + skip), rescaled.  fp32."""
    from storm_amd import ops
    g = golden["f1_ops"]
    P = {k[len(f"res_{tag}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"res_{tag}_")}
    d = lambda name: P[name].to(dev)
    dt = torch.float32
    x = nhwc(P["x"]).to(dev)
    Cout = P["Conv_0.weight"].shape[0]
    st0 = ops.gn_stats(x)
    if tag in ("up", "down"):
        h, xr = ops.gn_apply(x, st0, d("GroupNorm_0.weight"), d("GroupNorm_0.bias"), resample=1 if tag == "up" else 2)
    else:
        h, xr = ops.gn_apply(x, st0, d("GroupNorm_0.weight"), d("GroupNorm_0.bias")), x
    tb = ops.dense(F.silu(P["temb"]).to(dev), d("Dense_0.weight"), d("Dense_0.bias"))
    h, part = ops.conv([ops.Seg(h, ops.pack_conv_weight(d("Conv_0.weight"), dt), 9)], Cout, bias=d("Conv_0.bias"), tbias=tb,
                       gn_partials=True)
    st1 = ops.gn_finalize(part)
    ref1 = ops.gn_stats(h)
    assert torch.allclose(st1.cpu(), ref1.cpu(), rtol=1e-5, atol=1e-5 * float(ref1.abs().max()))
    h = ops.gn_apply(h, st1, d("GroupNorm_1.weight"), d("GroupNorm_1.bias"))
    segs = [ops.Seg(h, ops.pack_conv_weight(d("Conv_1.weight"), dt), 9)]
    if "Conv_2.weight" in P:
        segs.append(ops.Seg(xr, ops.pack_conv_weight(d("Conv_2.weight"), dt), 1))
        y = ops.conv(segs, Cout, bias=d("Conv_1.bias") + d("Conv_2.bias"), scale=2 ** -0.5)
    else:
        y = ops.conv(segs, Cout, bias=d("Conv_1.bias"), skip=xr, scale=2 ** -0.5)
    assert rel_l2(nchw(y.cpu()), P["y"]) < 5e-6


def test_attnblock_vs_reference_golden(dev, golden):
    """AttnBlockpp (layerspp.py:60-91) against the REFERENCE's output (fixture F1 attn_*), fp32, op-by-op path of the
    planner: GroupNorm -> NIN q, k, v (1x1 GEMMs) -> scores by the batched GEMM -> row softmax -> weights x v -> NIN_3 + skip."""
    from storm_amd import ops
    g = golden["f1_ops"]
    P = {k[len("attn_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("attn_")}
    d = lambda name: P[name].to(dev)
    dt = torch.float32
    B, C, H, W = P["x"].shape
    Lp = H * W
    x = nhwc(P["x"]).to(dev)
    h = ops.gn_apply(x, ops.gn_stats(x), d("GroupNorm_0.weight"), d("GroupNorm_0.bias"), silu=False).reshape(B, 1, Lp, C)
    Wm = [ops.pack_matrix(d(f"NIN_{i}.W"), dt, transpose=True) for i in range(4)]
    q = ops.conv([ops.Seg(h, Wm[0], 1)], C, bias=d("NIN_0.b"))
    k = ops.conv([ops.Seg(h, Wm[1], 1)], C, bias=d("NIN_1.b"))
    v = ops.conv([ops.Seg(h, Wm[2], 1)], C, bias=d("NIN_2.b"))
    # reference semantics through torch on the HIP tensors' VALUES would hide the kernels: use the row softmax kernel and
    # the batched-weights GEMM (scores[b] = q[b] k[b]^T: k as the per-batch weight matrix, rows padded to 32)
    LpP = ops.round_up(Lp, 32)
    kw = torch.zeros(B, LpP, C, device=dev); kw[:, :Lp] = k.reshape(B, Lp, C)
    s = ops.conv([ops.Seg(q, kw.reshape(B, 1, LpP, C), 1, w_batched=True)], Lp, outC=LpP, scale=C ** -0.5)
    w = ops.softmax_rows(s.reshape(B, Lp, LpP), dt, valid=Lp)
    vT = torch.zeros(B, ops.round_up(C, 32), LpP, device=dev); vT[:, :C, :Lp] = v.reshape(B, Lp, C).transpose(1, 2)
    o = ops.conv([ops.Seg(w.reshape(B, 1, Lp, LpP), vT.reshape(B, 1, -1, LpP), 1, w_batched=True)], C)
    y = ops.conv([ops.Seg(o, Wm[3], 1)], C, bias=d("NIN_3.b"), skip=x.reshape(B, 1, Lp, C), scale=2 ** -0.5)
    assert rel_l2(nchw(y.reshape(B, H, W, C).cpu()), P["y"]) < 5e-6


@pytest.mark.parametrize("fac", [0.15, 0.33])
def test_frontend_4s_vs_reference_golden(dev, golden, fac):
    """4 s utterance (64000 samples -> 500 frames padded to 512): spectrogram slice around the padded-frame boundary and the
    head / tail of the resynthesised waveform against the reference (fixture F5 L64000_*)."""
    from storm_amd import ops
    g = golden["f5_frontend"]
    key = f"L64000_f{int(fac * 100)}"
    y = torch.randn(1, 64000, generator=torch.Generator().manual_seed(1234 + 64000)) * 0.1
    yd = y.to(dev)
    Y = ops.stft(yd, ops.peak_abs(yd), spec_factor=fac, spec_abs_exponent=0.5, pad_to=64)
    assert Y.shape == (1, 256, 512)
    assert rel_l2(Y[..., 245:262].cpu(), torch.from_numpy(g[f"{key}_Yslice"])[0]) < 5e-6
    w = ops.istft(Y, 64000, None, spec_factor=fac, spec_abs_exponent=0.5).cpu()
    assert rel_l2(w[..., -512:], g[f"{key}_wavtail"]) < 5e-6
    assert rel_l2(w[..., :512], g[f"{key}_wavhead"]) < 5e-6


@pytest.mark.parametrize("e,fac", [(0.5, 0.15), (1.0, 1.0), (0.667, 0.065)])
def test_spec_transform_standalone(dev, e, fac):
    """storm_spec_transform == spec_fwd / spec_back (data_module.py:182-207) and they invert each other"""
    from storm_amd import ops
    from oracle import frontend_ref as FR
    g = torch.Generator().manual_seed(77)
    s = torch.complex(torch.randn(2, 256, 40, generator=g), torch.randn(2, 256, 40, generator=g))
    s[0, 0, 0] = 0                                            # |z| = 0: angle 0, stays 0
    f = ops.spec_transform(s.to(dev), fac, e, inverse=False)
    assert rel_l2(f.cpu(), FR.spec_fwd(s, fac, e)) < 2e-6
    b = ops.spec_transform(f, fac, e, inverse=True)
    assert rel_l2(b.cpu(), FR.spec_back(FR.spec_fwd(s, fac, e), fac, e)) < 2e-6
    assert rel_l2(b.cpu(), s) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shortcut,H,W,cin", [(0, 128, 256, 128), (128, 128, 256, 128), (0, 256, 512, 384), (0, 256, 512, 256)])
def test_conv_pipe128_bench_layer_vs_generic_kernel(shortcut, H, W, cin, switch):
    """The layers of BASELINE.json configs[1] the dispatcher gives to conv_pipe128.hip (128 -> 128 @ 128 x 256, batch 16, fused
    GroupNorm operand + temb bias + statistics epilogue, with / without the fused 1x1 shortcut; 384 -> 128 and 256 -> 128 @ 256 x 512)
    at FULL size: same output as the generic kernel up to the accumulation order (bf16 outputs: a few values differ by one rounding),
    same statistics partials."""
    from storm_amd import ops
    from tests.backend import setup_backend
    dev = setup_backend("hip")
    g = torch.Generator().manual_seed(7)
    B, cout, dt = 16, 128, torch.bfloat16
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(B, H, W, cin).to(dt).to(dev)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    segs = [ops.Seg(x, ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt), 9, gn_ss=ss, gn_silu=True)]
    if shortcut:
        segs.append(ops.Seg(rnd(B, H, W, shortcut).to(dt).to(dev), ops.pack_conv_weight((rnd(cout, shortcut, 1, 1) * 0.05).to(dev), dt), 1))
    kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), scale=0.7)
    assert ops.conv_kernel_name(segs, cout, **kw).startswith("storm::conv_pipe128_kernel")          # the production choice
    y, part = ops.conv(segs, cout, gn_partials=True, **kw)
    switch("STORM_CONV_PIPE128", 0)
    assert ops.conv_kernel_name(segs, cout, **kw).startswith("storm::conv_igemm_kernel")
    y0, part0 = ops.conv(segs, cout, gn_partials=True, **kw)
    assert rel_l2(y.float().cpu(), y0.float().cpu()) < 1e-3
    assert torch.allclose(part.cpu(), part0.cpu(), rtol=1e-3, atol=1e-3 * float(part0.abs().max()))
