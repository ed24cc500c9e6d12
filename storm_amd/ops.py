"""Tensor-level wrappers over the C ABI (one function per entry point of include/storm_hip.h).

Activations are NHWC tensors ``[B, H, W, C]`` (C % 8 == 0) in float32 or bfloat16; complex
spectrograms are complex64 ``[B, F, T]``.  These wrappers only marshal pointers/shapes and
enqueue on torch's current HIP stream; all compute happens in libstorm_hip.
"""
import ctypes as C
import math
import threading

import torch

from . import _lib as L


def _alloc(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def round_up(x, m):
    return (x + m - 1) // m * m


# ---------------------------------------------------------------- weights -----------------
def pack_conv_weight(w, dtype, cout_pad=32):
    """nn.Conv2d weight [Cout, Cin, kh, kw] fp32 -> packed [taps][CoutP][CinP]."""
    Cout, Cin, kh, kw = w.shape
    ntaps = kh * kw
    per16 = 4 if dtype == torch.float32 else 8
    CoutP, CinP = round_up(Cout, cout_pad), round_up(Cin, per16 * 2)
    w = w.contiguous().float()
    out = _alloc((ntaps, CoutP, CinP), dtype, w)
    L.check(L.lib().storm_pack_conv_weight(L.ptr(w), L.ptr(out), Cout, Cin, ntaps, CoutP, CinP, L.dt(dtype), L.stream()),
            "storm_pack_conv_weight")
    return out


def pack_matrix(w, dtype, transpose=False, cout_pad=32):
    """[Cout][Cin] fp32 (or [Cin][Cout] with transpose=True, e.g. NIN.W) -> packed [CoutP][CinP]."""
    w = w.contiguous().float()
    Cin, Cout = (w.shape[0], w.shape[1]) if transpose else (w.shape[1], w.shape[0])
    per16 = 4 if dtype == torch.float32 else 8
    CoutP, CinP = round_up(Cout, cout_pad), round_up(Cin, per16 * 2)
    out = _alloc((1, CoutP, CinP), dtype, w)
    L.check(L.lib().storm_pack_matrix(L.ptr(w), L.ptr(out), Cout, Cin, int(transpose), CoutP, CinP, L.dt(dtype), L.stream()),
            "storm_pack_matrix")
    return out


# ---------------------------------------------------------------- conv / gemm -------------
class Seg:
    """One K-segment of storm_conv: activations (optionally the channel concat of two tensors)
    and packed weights [ntaps][rows][CinP] (or a per-batch activation used as the weight matrix)."""

    def __init__(self, src_a, w, ntaps, src_b=None, w_batched=False, src_bstride=None, gn_ss=None, gn_silu=True):
        self.src_a, self.src_b, self.w, self.ntaps = src_a, src_b, w, ntaps
        self.w_batched, self.src_bstride = w_batched, src_bstride
        self.gn_ss, self.gn_silu = gn_ss, gn_silu      # fused GroupNorm apply (+SiLU) on load


def _conv_args(segs, Cout, out=None, outC=None, bias=None, tbias=None, skip=None, scale=1.0, out_f32=False,
               B=None, H=None, W=None):
    """storm_conv_args for a convolution (+ the output tensor it points at)"""
    x = segs[0].src_a
    if B is None:
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
    dtype = x.dtype
    outC = outC if outC is not None else round_up(Cout, 8)
    if out is None:
        out = _alloc((B, H, W, outC), torch.float32 if out_f32 else dtype, x)
    a = L.ConvArgs()
    a.nseg = len(segs)
    for k, s in enumerate(segs):
        g = a.seg[k]
        g.src_a, g.Ca = L.ptr(s.src_a), s.src_a.shape[-1]
        g.bstride_a = s.src_bstride if s.src_bstride is not None else H * W * g.Ca
        if s.src_b is not None:
            g.src_b, g.Cb = L.ptr(s.src_b), s.src_b.shape[-1]
            g.bstride_b = H * W * g.Cb
        g.w, g.ntaps = L.ptr(s.w), s.ntaps
        g.CinP, g.w_rows = s.w.shape[-1], s.w.shape[-2]
        g.w_tapstride = s.w.shape[-1] * s.w.shape[-2]
        g.w_bstride = s.w.shape[-1] * s.w.shape[-2] if s.w_batched else 0
        if s.gn_ss is not None:
            g.gn_ss, g.gn_silu = L.ptr(s.gn_ss), int(s.gn_silu)
    a.B, a.H, a.W = B, H, W
    a.out, a.outC, a.Cout, a.out_bstride = L.ptr(out), outC, Cout, H * W * outC
    a.bias = L.ptr(bias)
    if tbias is not None:
        a.tbias, a.tbias_stride = L.ptr_rows(tbias), tbias.stride(0)
    if skip is not None:
        a.skip, a.skip_bstride = L.ptr(skip), H * W * outC
    a.scale, a.out_f32, a.dtype = scale, int(out_f32), L.dt(dtype)
    return a, out


def conv(segs, Cout, gn_partials=False, **kw):
    """out[b,h,w,co] = (sum_seg conv(seg) + bias + tbias[b] + skip) * scale   (storm_conv)."""
    a, out = _conv_args(segs, Cout, **kw)
    part = None
    if gn_partials:
        tiles = L.lib().storm_conv_tiles(C.byref(a))
        part = torch.zeros((a.B, tiles, a.outC, 2), dtype=torch.float32, device=out.device)
        a.gn_part = L.ptr(part)
    need = L.lib().storm_conv_splitk_bytes(C.byref(a))      # few-tile 3x3 layers: scratch for the split of K over workgroups
    if need > 0:
        ws = torch.empty((need,), dtype=torch.uint8, device=out.device)
        a.splitk_ws, a.splitk_ws_bytes = L.ptr(ws), need
    L.check(L.lib().storm_conv(C.byref(a), L.stream()), "storm_conv")
    return (out, part) if gn_partials else out


def conv_group(problems, Cout, gn_partials=False, bn=0):
    """ONE launch for several problems of one 3x3 layer (storm_conv_group): problems = [(segs, kw), ...] as conv() takes them, every
    problem with its own tensors / batch size / width and the same weights.  Returns the outputs (and GroupNorm partials) per problem."""
    P = len(problems)
    arr = (L.ConvArgs * P)()
    outs, parts, keep = [], [], []
    for p, (segs, kw) in enumerate(problems):
        a, out = _conv_args(segs, Cout, **kw)
        part = None
        if gn_partials:
            tiles = L.lib().storm_conv_tiles(C.byref(a))
            part = torch.zeros((a.B, tiles, a.outC, 2), dtype=torch.float32, device=out.device)
            a.gn_part = L.ptr(part)
        arr[p] = a
        keep.append(a)
        outs.append(out)
        parts.append(part)
    need = L.lib().storm_conv_group_blob_bytes(arr, P)
    blob = torch.empty((need,), dtype=torch.uint8, device=outs[0].device)
    L.check(L.lib().storm_conv_group(arr, P, L.ptr(blob), need, int(bn), L.stream()), "storm_conv_group")
    return (outs, parts) if gn_partials else outs


def conv_kernel_name(segs, Cout, **kw):
    """name of the kernel storm_conv launches for these arguments (storm_conv_kernel_name)"""
    a, _ = _conv_args(segs, Cout, **kw)
    if L.lib().storm_conv_splitk_bytes(C.byref(a)) > 0:     # (as conv() would call it: with the split-K scratch)
        a.splitk_ws, a.splitk_ws_bytes = 16, L.lib().storm_conv_splitk_bytes(C.byref(a))
    return L.lib().storm_conv_kernel_name(C.byref(a)).decode()


def gn_finalize(part_a, part_b=None, gamma=None, beta=None, count=None, eps=1e-6):
    """[B][tiles][C][2] conv-epilogue partials (optionally of two concatenated tensors) -> stats [B][G][2] fp64;
    with gamma/beta/count also the per-channel affine for convs that fuse the apply, ss [B][C/8][2][8] (per channel octet:
    its 8 scales, then its 8 shifts - see pack_gn_ss)."""
    B, ta, Ca, _ = part_a.shape
    tb, Cb = (part_b.shape[1], part_b.shape[2]) if part_b is not None else (0, 0)
    G = gn_groups(Ca + Cb)
    stats = torch.empty((B, G, 2), dtype=torch.float64, device=part_a.device)
    if gamma is None:
        L.check(L.lib().storm_gn_finalize(L.ptr(part_a), Ca, ta, L.ptr(part_b), Cb, tb, B, G, L.ptr(stats), L.stream()),
                "storm_gn_finalize")
        return stats
    ss = torch.empty((B, (Ca + Cb) // 8, 2, 8), dtype=torch.float32, device=part_a.device)
    L.check(L.lib().storm_gn_finalize_ss(L.ptr(part_a), Ca, ta, L.ptr(part_b), Cb, tb, B, G, int(count), L.ptr(gamma),
                                         L.ptr(beta), eps, L.ptr(stats), L.ptr(ss), L.stream()), "storm_gn_finalize_ss")
    return stats, ss


def pack_gn_ss(scale, shift):
    """per-channel (scale, shift) [B, C] -> the table layout the conv kernels read (storm_gn_finalize_ss)"""
    B, Cc = scale.shape
    return torch.stack([scale.reshape(B, Cc // 8, 8), shift.reshape(B, Cc // 8, 8)], 2).contiguous().float()


def attention(q, k, vT, bias, scale):
    """q, k [B, L, C], vT [B, C, ldv] (bf16) -> softmax(scale q k^T) v + bias, [B, L, C] (storm_attention)."""
    B, Lq, Cc = q.shape
    out = torch.empty_like(q)
    need = L.lib().storm_attention_scratch_bytes(B, Lq, Cc, L.dt(q))        # small calls: scratch for the split of the key loop
    ws = torch.empty((need,), dtype=torch.uint8, device=q.device) if need > 0 else None
    L.check(L.lib().storm_attention_ws(L.ptr(q), L.ptr(k), L.ptr(vT), L.ptr(bias), L.ptr(out), B, Lq, Cc, vT.shape[-1], Lq * Cc, Lq * Cc,
                                       Cc * vT.shape[-1], Lq * Cc, float(scale), L.dt(q), L.ptr(ws), need, L.stream()), "storm_attention")
    return out


def attention_group(problems, bias, scale):
    """ONE launch for the fused attention of several problems of one layer (storm_attention_group): problems = [(q, k, vT), ...] as
    attention() takes them (own batch sizes / sequence lengths).  Returns the outputs per problem."""
    P = len(problems)
    Cc = problems[0][0].shape[-1]
    arr = lambda vals: (C.c_void_p * P)(*vals)
    outs = [torch.empty_like(q) for q, _, _ in problems]
    Bs = (C.c_int * P)(*[q.shape[0] for q, _, _ in problems])
    Ls = (C.c_int * P)(*[q.shape[1] for q, _, _ in problems])
    ld = (C.c_int * P)(*[v.shape[-1] for _, _, v in problems])
    need = L.lib().storm_attention_group_blob_bytes(Bs, Ls, P)
    blob = torch.empty((need,), dtype=torch.uint8, device=outs[0].device)
    L.check(L.lib().storm_attention_group(arr([L.ptr(q) for q, _, _ in problems]), arr([L.ptr(k) for _, k, _ in problems]),
                                          arr([L.ptr(v) for _, _, v in problems]), arr([L.ptr(o) for o in outs]), Bs, Ls, ld, P, L.ptr(bias), Cc,
                                          float(scale), L.dt(problems[0][0]), L.ptr(blob), need, L.stream()), "storm_attention_group")
    return outs


# ---------------------------------------------------------------- norm / resample ---------
def gn_groups(C):
    return min(C // 4, 32)


def gn_stats(xa, xb=None):
    B, H, W, Ca = xa.shape
    Cb = xb.shape[-1] if xb is not None else 0
    G = gn_groups(Ca + Cb)
    stats = torch.zeros((B, G, 2), dtype=torch.float64, device=xa.device)
    L.check(L.lib().storm_gn_stats(L.ptr(xa), Ca, L.ptr(xb), Cb, B, H * W, G, L.ptr(stats), L.dt(xa), L.stream()),
            "storm_gn_stats")
    return stats


def gn_apply(xa, stats, gamma, beta, xb=None, silu=True, resample=0, eps=1e-6, want_raw=True):
    B, H, W, Ca = xa.shape
    Cb = xb.shape[-1] if xb is not None else 0
    Ctot = Ca + Cb
    OH, OW = (2 * H, 2 * W) if resample == 1 else ((H // 2, W // 2) if resample == 2 else (H, W))
    out = _alloc((B, OH, OW, Ctot), xa.dtype, xa)
    raw = _alloc((B, OH, OW, Ctot), xa.dtype, xa) if (resample and want_raw) else None
    L.check(L.lib().storm_gn_apply(L.ptr(xa), Ca, L.ptr(xb), Cb, B, H, W, stats.shape[1], L.ptr(stats),
                                   L.ptr(gamma), L.ptr(beta), eps, int(silu), resample, L.ptr(out), L.ptr(raw),
                                   L.dt(xa), L.stream()), "storm_gn_apply")
    return (out, raw) if resample else out


def fir_up2(x, add=None):
    B, H, W, Cc = x.shape
    out = _alloc((B, 2 * H, 2 * W, Cc), x.dtype, x)
    L.check(L.lib().storm_fir_up2(L.ptr(x), L.ptr(add), L.ptr(out), B, H, W, Cc, L.dt(x), L.stream()), "storm_fir_up2")
    return out


def fir_down2(x):
    B, H, W, Cc = x.shape
    out = _alloc((B, H // 2, W // 2, Cc), x.dtype, x)
    L.check(L.lib().storm_fir_down2(L.ptr(x), L.ptr(out), B, H, W, Cc, L.dt(x), L.stream()), "storm_fir_down2")
    return out


def upfirdn2d(input, kernel, up_x=1, up_y=1, down_x=1, down_y=1, pad_x0=0, pad_x1=0, pad_y0=0, pad_y1=0):
    """The reference's native op with its own argument list (op/upfirdn2d.cpp:12-22): input [N, H, W, 1] contiguous (planes),
    kernel [kh, kw] -> [N, outH, outW, 1]; any factors / pads.  One call of storm_upfirdn2d on torch's current stream."""
    if input.dim() != 4 or input.shape[-1] != 1 or not input.is_contiguous():
        raise ValueError("upfirdn2d: input must be a contiguous [N, H, W, 1] tensor (op/upfirdn2d.cpp:9)")
    N, H, W, _ = input.shape
    kh, kw = kernel.shape
    lib = L.lib()
    OH = lib.storm_upfirdn2d_out_size(H, up_y, down_y, pad_y0, pad_y1, kh)
    OW = lib.storm_upfirdn2d_out_size(W, up_x, down_x, pad_x0, pad_x1, kw)
    out = _alloc((N, max(OH, 0), max(OW, 0), 1), input.dtype, input)
    k32 = kernel.to(device=input.device, dtype=torch.float32).contiguous()
    L.check(lib.storm_upfirdn2d(L.ptr(input), L.ptr(k32), L.ptr(out), N, H, W, kh, kw, up_x, up_y, down_x, down_y,
                                pad_x0, pad_x1, pad_y0, pad_y1, L.dt(input), L.stream()), "storm_upfirdn2d")
    return out


def softmax_rows(scores, dtype, valid=None):
    """softmax over the first `valid` columns of each row (the rest is row padding, written as 0)."""
    ld = scores.shape[-1]
    rows, Lr = scores.numel() // ld, (valid if valid is not None else ld)
    out = _alloc(scores.shape, dtype, scores)
    L.check(L.lib().storm_softmax_rows(L.ptr(scores), L.ptr(out), rows, Lr, ld, L.dt(dtype), L.stream()), "storm_softmax_rows")
    return out


# ---------------------------------------------------------------- network head/tail -------
def pack_input(cplx, dtype):
    """list of complex64 [B,F,T] -> NHWC [B,F,T,8] holding 2*(re,im)-1."""
    B, F, T = cplx[0].shape
    views = [torch.view_as_real(c.contiguous()) for c in cplx]
    arr = (C.c_void_p * len(views))(*[L.ptr(v) for v in views])
    out = _alloc((B, F, T, 8), dtype, views[0])
    L.check(L.lib().storm_pack_input(arr, len(views), L.ptr(out), B, F, T, L.dt(dtype), L.stream()), "storm_pack_input")
    return out


def time_embedding(t, gfp_W, W1, b1, W2, b2):
    B, nf = t.shape[0], gfp_W.shape[0]
    out = _alloc((B, 4 * nf), torch.float32, t)
    L.check(L.lib().storm_time_embedding(L.ptr(t), L.ptr(gfp_W), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2),
                                         L.ptr(out), B, nf, L.stream()), "storm_time_embedding")
    return out


def dense(x, W, bias):
    B, K = x.shape
    N = W.shape[0]
    out = _alloc((B, N), torch.float32, x)
    L.check(L.lib().storm_dense(L.ptr(x), L.ptr(W), L.ptr(bias), L.ptr(out), B, N, K, L.stream()), "storm_dense")
    return out


def output_head(pyr, t, W, bias, negate):
    B, F, T, _ = pyr.shape
    cin = W.shape[1]
    out = torch.empty((B, F, T), dtype=torch.complex64, device=pyr.device)
    W2 = W.reshape(2, cin).contiguous()
    L.check(L.lib().storm_output_head(L.ptr(pyr), L.ptr(t), L.ptr(W2), L.ptr(bias), cin,
                                      L.ptr(torch.view_as_real(out)), B, F, T, int(negate), L.dt(pyr), L.stream()),
            "storm_output_head")
    return out


def input_pyramid(cplx, dtype, n_levels, level0=None):
    """The progressive input pyramid in ONE launch (storm_input_pyramid): level 0 = pack_input(cplx), level k = fir_down2 of level k - 1.
    cplx None: `level0` (NHWC [B,F,T,8]) is read instead.  Returns the list of levels."""
    if cplx is not None:
        B, F, T = cplx[0].shape
        views = [torch.view_as_real(c.contiguous()) for c in cplx]
        arr = (C.c_void_p * len(views))(*[L.ptr(v) for v in views])
        like, n_in = views[0], len(views)
        levels = [_alloc((B, F, T, 8), dtype, like)]
    else:
        B, F, T, _ = level0.shape
        arr, like, n_in = None, level0, 0
        levels = [level0]
    levels += [_alloc((B, F >> k, T >> k, 8), dtype, like) for k in range(1, n_levels)]
    lv = (C.c_void_p * n_levels)(*[L.ptr(v) for v in levels])
    L.check(L.lib().storm_input_pyramid(arr, n_in, lv, n_levels, B, F, T, L.dt(dtype), L.stream()), "storm_input_pyramid")
    return levels


def output_pyramid(phs, t, W, bias, negate):
    """The progressive output pyramid + head in ONE launch (storm_output_pyramid): phs = the pyramid convolutions' outputs NHWC [B, F >> k, T >> k, 8],
    finest first; p = phs[0] + up(phs[1] + up(...)); returns complex64 [B,F,T] = output_head(p, t, W, bias, negate)."""
    B, F, T, _ = phs[0].shape
    cin = W.shape[1]
    out = torch.empty((B, F, T), dtype=torch.complex64, device=phs[0].device)
    W2 = W.reshape(2, cin).contiguous()
    arr = (C.c_void_p * len(phs))(*[L.ptr(v) for v in phs])
    L.check(L.lib().storm_output_pyramid(arr, len(phs), L.ptr(t), L.ptr(W2), L.ptr(bias), cin, L.ptr(torch.view_as_real(out)), B, F, T, int(negate),
                                         L.dt(phs[0]), L.stream()), "storm_output_pyramid")
    return out


# ---------------------------------------------------------------- SDE steps ---------------
def _ouve(sde):
    return L.Ouve(float(sde.theta), float(sde.sigma_min), float(sde.sigma_max), int(sde.N))


def _r(z):
    return None if z is None else torch.view_as_real(z)


def _t32(t):
    """per-batch times as the kernels read them: contiguous float32 (a float64 / strided t would be read as garbage)"""
    return t.to(torch.float32).contiguous()


def _n_per_batch(x):
    return x.numel() // x.shape[0]


def ouve_prior(sde, y, z=None, seed=0, offset=0):
    x = torch.empty_like(y)
    L.check(L.lib().storm_ouve_prior(L.ptr(_r(y)), L.ptr(_r(z)), L.ptr(_r(x)), y.shape[0], _n_per_batch(y), _ouve(sde),
                                     seed, offset, L.stream()), "storm_ouve_prior")
    return x


def ouve_ald_step(sde, x, score, t, snr, z=None, seed=0, offset=0):
    """In place on x; returns (x, x_mean)."""
    xm = torch.empty_like(x)
    t = _t32(t)
    L.check(L.lib().storm_ouve_ald_step(L.ptr(_r(x)), L.ptr(_r(xm)), L.ptr(_r(score)), L.ptr(_r(z)), L.ptr(t),
                                        x.shape[0], _n_per_batch(x), _ouve(sde), float(snr), seed, offset, L.stream()),
            "storm_ouve_ald_step")
    return x, xm


def ouve_predictor_step(sde, x, score, y, t, kind=0, z=None, noise_free=False, seed=0, offset=0):
    """In place on x; returns (x, x_mean)."""
    xm = torch.empty_like(x)
    t = _t32(t)
    L.check(L.lib().storm_ouve_predictor_step(L.ptr(_r(x)), L.ptr(_r(xm)), L.ptr(_r(score)), L.ptr(_r(y)), L.ptr(_r(z)),
                                              L.ptr(t), x.shape[0], _n_per_batch(x), _ouve(sde), kind, int(noise_free),
                                              seed, offset, L.stream()), "storm_ouve_predictor_step")
    return x, xm


def batch_l2norm(v):
    out = _alloc((v.shape[0],), torch.float32, v)
    L.check(L.lib().storm_batch_l2norm(L.ptr(_r(v)), L.ptr(out), v.shape[0], _n_per_batch(v), L.stream()), "storm_batch_l2norm")
    return out


def _all_reduce_sum(t, group):
    """in-place SUM of `t` over a torch.distributed ProcessGroup (RCCL on HIP devices, gloo on CPU tensors).  A ProcessGroup
    has no `all_reduce` method (its `allreduce` returns a Work handle): the functional form is the supported call."""
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def langevin_group_norms(sn, zn, group):
    """(mean ||score||, mean ||z||) over the rows of ALL ranks' batches: the two floats (+ the row count) a sharded `langevin`
    corrector needs to reproduce the unsharded step size (correctors.py:53-55)"""
    tot = torch.stack([sn.sum(), zn.sum(), torch.tensor(float(sn.shape[0]), device=sn.device, dtype=sn.dtype)])
    _all_reduce_sum(tot, group)
    return (tot[0:1] / tot[2]).contiguous(), (tot[1:2] / tot[2]).contiguous()


def langevin_step(x, score, z, snr, per_row=False, group=None):
    """Langevin corrector update, IN PLACE on x (returns (x, x_mean)).  Step size from the batch-mean norms (reference
    semantics, correctors.py:53-55), from every row's own norms (per_row: B independent batch-1 calls), or - with a
    torch.distributed process group - from the means over the batches of all ranks (a 2-float all-reduce per step)."""
    xm = torch.empty_like(x)
    sn, zn = batch_l2norm(score), batch_l2norm(z)
    mode = 1 if per_row else 0
    if group is not None and not per_row:
        (sn, zn), mode = langevin_group_norms(sn, zn, group), 2
    L.check(L.lib().storm_langevin_step(L.ptr(_r(x)), L.ptr(_r(xm)), L.ptr(_r(score)), L.ptr(_r(z)), L.ptr(sn), L.ptr(zn),
                                        x.shape[0], _n_per_batch(x), float(snr), mode, L.stream()), "storm_langevin_step")
    return x, xm


def si_sdr(s, s_hat, eps=0.0):
    """SI-SDR in dB per row of two fp32 waveform batches [B, L] (rows may be strided views)."""
    assert s.dim() == 2 and s_hat.dim() == 2 and s.shape[0] == s_hat.shape[0] and s.stride(1) == 1 and s_hat.stride(1) == 1
    n = min(s.shape[1], s_hat.shape[1])
    out = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
    L.check(L.lib().storm_si_sdr(L.ptr_rows(s), L.ptr_rows(s_hat), L.ptr(out), s.shape[0], n, s.stride(0), s_hat.stride(0), float(eps),
                                 L.stream()), "storm_si_sdr")
    return out


def ouve_pf_drift(sde, x, y, score, t):
    """theta (y - x) - 1/2 g(t)^2 score: the right-hand side of the probability-flow ODE in one pass."""
    out = torch.empty_like(x)
    t = _t32(t)                                             # (kept alive in a local: a temporary would be freed before the launch)
    L.check(L.lib().storm_ouve_pf_drift(L.ptr(_r(out)), L.ptr(_r(x)), L.ptr(_r(y)), L.ptr(_r(score)), L.ptr(t), x.shape[0],
                                        _n_per_batch(x), _ouve(sde), L.stream()), "storm_ouve_pf_drift")
    return out


def ouve_pf_drift_g(sde, x, y, score, g_rows):
    """theta (y - x) - 1/2 g_b^2 score with the diffusion coefficient given per row (device fp32 [B])"""
    out = torch.empty_like(x)
    g_rows = g_rows.to(device=x.device, dtype=torch.float32).contiguous()
    L.check(L.lib().storm_ouve_pf_drift_g(L.ptr(_r(out)), L.ptr(_r(x)), L.ptr(_r(y)), L.ptr(_r(score)), L.ptr(g_rows), x.shape[0],
                                          _n_per_batch(x), float(sde.theta), L.stream()), "storm_ouve_pf_drift_g")
    return out


def _rows32(v, like):
    """per-row coefficients as the rows-form kernels read them: device fp32 [B], contiguous"""
    return v.to(device=like.device, dtype=torch.float32).contiguous()


def sde_prior_rows(y, std_rows, z=None, seed=0, offset=0):
    """y + z * std_b (OUVPSDE.prior_sampling, sdes.py:306-310); z=None draws in-kernel (Philox)."""
    x = torch.empty_like(y)
    std_rows = _rows32(std_rows, y)
    L.check(L.lib().storm_sde_prior_rows(L.ptr(_r(y)), L.ptr(_r(z)), L.ptr(_r(x)), L.ptr(std_rows), y.shape[0], _n_per_batch(y),
                                         seed, offset, L.stream()), "storm_sde_prior_rows")
    return x


def sde_predictor_step_rows(sde, x, score, y, t, kind=0, z=None, noise_free=False, seed=0, offset=0):
    """Predictor update for an SDE with drift a(t) (y - x) and diffusion g(t) given by its `drift_rows(t)` / `diffusion(t)`
    (fp32 [B], the reference's own expressions).  In place on x; returns (x, x_mean)."""
    xm = torch.empty_like(x)
    a, g = _rows32(sde.drift_rows(t), x), _rows32(sde.diffusion(t), x)
    L.check(L.lib().storm_sde_predictor_step_rows(L.ptr(_r(x)), L.ptr(_r(xm)), L.ptr(_r(score)), L.ptr(_r(y)), L.ptr(_r(z)), L.ptr(a),
                                                  L.ptr(g), x.shape[0], _n_per_batch(x), int(sde.N), kind, int(noise_free), seed,
                                                  offset, L.stream()), "storm_sde_predictor_step_rows")
    return x, xm


def sde_pf_drift_rows(x, y, score, a_rows, g_rows):
    """a_b (y - x) - 1/2 g_b^2 score: the probability-flow right-hand side with per-row coefficients (device fp32 [B])"""
    out = torch.empty_like(x)
    a, g = _rows32(a_rows, x), _rows32(g_rows, x)
    L.check(L.lib().storm_sde_pf_drift_rows(L.ptr(_r(out)), L.ptr(_r(x)), L.ptr(_r(y)), L.ptr(_r(score)), L.ptr(a), L.ptr(g),
                                            x.shape[0], _n_per_batch(x), L.stream()), "storm_sde_pf_drift_rows")
    return out


_rk_scratch = {}


def _kptrs(K):
    arr = (C.c_void_p * len(K))(*[L.ptr(_r(k)) for k in K])
    return arr


def rk_combine(x, K, coef, h, out=None):
    """out = x + h * sum_j coef[j] K[j] (one fused pass; complex64 tensors of one shape)."""
    out = torch.empty_like(x) if out is None else out
    cf = (C.c_float * len(K))(*[float(c) for c in coef])
    L.check(L.lib().storm_rk_combine(L.ptr(_r(out)), L.ptr(_r(x)), _kptrs(K), cf, len(K), float(h), x.numel(), L.stream()),
            "storm_rk_combine")
    return out


def rk_scaled_sumsq(xa, xb, K, coef, h, atol, rtol, mode=None):
    """Device scalar (float64 tensor [1]) = sum over complex elements of |v|^2 / (atol + max(|xa|, |xb|) rtol)^2 with
    v = h sum coef[j] K[j] (mode None), K[0] (mode -1) or K[0] - K[1] (mode -2); nothing is synchronised here."""
    key = (str(xa.device), threading.get_ident())       # (per host thread: grouped micro-batches interleave their launches on one stream)
    if key not in _rk_scratch:
        _rk_scratch[key] = torch.empty(2048, dtype=torch.float64, device=xa.device)
    out = torch.empty(1, dtype=torch.float64, device=xa.device)
    n_terms = len(K) if mode is None else mode
    cf = (C.c_float * max(1, len(K)))(*[float(c) for c in (coef if coef is not None else [0.0] * len(K))])
    L.check(L.lib().storm_rk_scaled_sumsq(L.ptr(out), L.ptr(_rk_scratch[key]), 2048, L.ptr(_r(xa)), L.ptr(_r(xb)) if xb is not None else None,
                                          _kptrs(K), cf, n_terms, float(h), float(atol), float(rtol), xa.numel(), L.stream()),
            "storm_rk_scaled_sumsq")
    return out


RK_MAX_ROWS, RK_ROW_BLOCKS = 128, 256          # include/storm_hip.h: STORM_RK_MAX_ROWS, STORM_RK_ROW_BLOCKS


def _hrows(h, B):
    if h is None:
        return None
    if len(h) != B or B > RK_MAX_ROWS:
        raise ValueError(f"per-row step sizes: {len(h)} values for {B} rows (at most {RK_MAX_ROWS})")
    return (C.c_double * B)(*[float(v) for v in h])


def rk_combine_rows(x64, K, coef, h_rows, want64=False):
    """x64[b] + h_rows[b] * sum_j coef[j] K[j][b]: one Runge-Kutta stage of B utterances with their own step sizes.  x64 is
    the complex128 solver state, K complex64 stages; returns the complex64 rounding (the next network input), with
    want64=True (out64, out32)."""
    B = x64.shape[0]
    if B > RK_MAX_ROWS:                      # the kernels take their step sizes as a 128-entry argument array: larger batches in row blocks
        parts = [rk_combine_rows(x64[i:i + RK_MAX_ROWS], [k[i:i + RK_MAX_ROWS] for k in K], coef, h_rows[i:i + RK_MAX_ROWS], want64)
                 for i in range(0, B, RK_MAX_ROWS)]
        return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])) if want64 else torch.cat(parts)
    out32 = torch.empty(x64.shape, dtype=torch.complex64, device=x64.device)
    out64 = torch.empty_like(x64) if want64 else None
    cf = (C.c_double * len(K))(*[float(c) for c in coef])
    L.check(L.lib().storm_rk_combine_rows(L.ptr(_r(out64)) if want64 else None, L.ptr(_r(out32)), L.ptr(_r(x64)), _kptrs(K), cf, len(K),
                                          _hrows(h_rows, B), B, _n_per_batch(x64), L.stream()), "storm_rk_combine_rows")
    return (out64, out32) if want64 else out32


def rk_scaled_sumsq_rows(xa, xb, K, coef, h_rows, atol, rtol, mode=None):
    """Device float64 [B]: row b's sum over its complex elements of |v|^2 / (atol + max(|xa|, |xb|) rtol)^2 with
    v = h_b sum coef[j] K[j] (mode None), K[0] (-1), K[0] - K[1] (-2) or xa itself (-3); xa / xb complex128, K complex64.
    A row's value does not depend on the other rows of the batch."""
    B = xa.shape[0]
    if B > RK_MAX_ROWS:                      # (row blocks, as rk_combine_rows: a row's value does not depend on the others)
        return torch.cat([rk_scaled_sumsq_rows(xa[i:i + RK_MAX_ROWS], xb[i:i + RK_MAX_ROWS] if xb is not None else None,
                                               [k[i:i + RK_MAX_ROWS] for k in K], coef,
                                               h_rows[i:i + RK_MAX_ROWS] if h_rows is not None else None, atol, rtol, mode)
                          for i in range(0, B, RK_MAX_ROWS)])
    key = (str(xa.device), "rows", threading.get_ident())
    need = RK_ROW_BLOCKS * B
    if key not in _rk_scratch or _rk_scratch[key].numel() < need:
        _rk_scratch[key] = torch.empty(need, dtype=torch.float64, device=xa.device)
    out = torch.empty(B, dtype=torch.float64, device=xa.device)
    n_terms = len(K) if mode is None else mode
    cf = (C.c_double * max(1, len(K)))(*[float(c) for c in (coef if coef is not None else [0.0] * max(1, len(K)))])
    L.check(L.lib().storm_rk_scaled_sumsq_rows(L.ptr(out), L.ptr(_rk_scratch[key]), _rk_scratch[key].numel(), L.ptr(_r(xa)),
                                               L.ptr(_r(xb)) if xb is not None else None, _kptrs(K) if K else None, cf, n_terms,
                                               _hrows(h_rows, B), float(atol), float(rtol), B, _n_per_batch(xa), L.stream()),
            "storm_rk_scaled_sumsq_rows")
    return out


def copy_rows(dst, src, mask):
    """dst[b] = src[b] for the rows with mask[b] (host booleans): the rows whose Runge-Kutta step was accepted"""
    B = dst.shape[0]
    m = (C.c_int * B)(*[int(bool(v)) for v in mask])
    L.check(L.lib().storm_copy_rows(L.ptr(dst), L.ptr(src), m, B, dst[0].numel() * dst.element_size(), L.stream()), "storm_copy_rows")
    return dst


def complex_randn(shape, device, seed, offset):
    z = torch.empty(shape, dtype=torch.complex64, device=device)
    L.check(L.lib().storm_complex_randn(L.ptr(_r(z)), z.numel(), seed, offset, L.stream()), "storm_complex_randn")
    return z


# ---------------------------------------------------------------- spectral ----------------
_tables = {}


def dft_tables(n_fft, device, window="hann"):
    """(window [n_fft], twiddle [n_fft,2] = (cos, sin)(2 pi k / n_fft)) on `device`, cached."""
    key = (n_fft, str(device), window)
    if key not in _tables:
        if window == "hann":
            w = torch.hann_window(n_fft, periodic=True)
        elif window == "sqrthann":
            w = torch.sqrt(torch.hann_window(n_fft, periodic=True))
        else:
            raise NotImplementedError(f"Window type {window} not implemented!")
        k = torch.arange(n_fft, dtype=torch.float64) * (2.0 * math.pi / n_fft)
        tw = torch.stack([torch.cos(k), torch.sin(k)], dim=1).float().contiguous()
        _tables[key] = (w.to(device).contiguous(), tw.to(device))
    return _tables[key]


def spec_transform(spec, spec_factor, spec_abs_exponent, inverse):
    """spec_fwd (inverse=False) / spec_back (inverse=True) on a complex64 tensor."""
    spec = spec.contiguous()
    out = torch.empty_like(spec)
    L.check(L.lib().storm_spec_transform(L.ptr(_r(spec)), L.ptr(_r(out)), spec.numel(), float(spec_factor),
                                         float(spec_abs_exponent), int(inverse), L.stream()), "storm_spec_transform")
    return out


def _row_len(lengths, like):
    """per-row sample counts of a ragged batch as the kernels read them (device int32 [B]) or None"""
    if lengths is None:
        return None
    if isinstance(lengths, torch.Tensor):
        return lengths.to(device=like.device, dtype=torch.int32).contiguous()
    return torch.tensor([int(v) for v in lengths], dtype=torch.int32, device=like.device)


def peak_abs(wav, lengths=None):
    B, Lw = wav.shape
    out = _alloc((B,), torch.float32, wav)
    rl = _row_len(lengths, wav)
    L.check(L.lib().storm_peak_abs(L.ptr(wav), L.ptr(out), B, Lw, wav.stride(0), L.ptr(rl), L.stream()), "storm_peak_abs")
    return out


def stft(wav, peak=None, n_fft=510, hop=128, spec_factor=1.0, spec_abs_exponent=1.0, pad_to=1, window="hann", lengths=None):
    """wav [B, L] fp32 -> complex64 [B, n_fft/2+1, Tpad] = spec_fwd(stft(wav / peak)), zero padded
    in T to a multiple of `pad_to`.  lengths: per-row sample counts of a ragged batch (rows zero filled past them)."""
    B, Lw = wav.shape
    # reflect padding by n_fft // 2 samples needs more samples than that in EVERY row (torch.stft raises otherwise,
    # data_module.py:217-219): a shorter row of a ragged batch would index before its own start
    shortest = Lw if lengths is None else int(min(int(v) for v in lengths))
    if shortest <= n_fft // 2:
        raise ValueError(f"stft: a row of {shortest} samples is too short for reflect padding by n_fft // 2 = {n_fft // 2}")
    if lengths is not None and int(max(int(v) for v in lengths)) > Lw:
        raise ValueError(f"stft: a row length exceeds the batch width {Lw}")
    n_frames = 1 + Lw // hop
    Tpad = round_up(n_frames, pad_to)
    win, tw = dft_tables(n_fft, wav.device, window)
    spec = torch.empty((B, n_fft // 2 + 1, Tpad), dtype=torch.complex64, device=wav.device)
    rl = _row_len(lengths, wav)
    L.check(L.lib().storm_stft(L.ptr(wav), L.ptr(peak), L.ptr(_r(spec)), L.ptr(win), L.ptr(tw), B, Lw, wav.stride(0),
                               n_fft, hop, n_frames, Tpad, float(spec_factor), float(spec_abs_exponent), L.ptr(rl), L.stream()),
            "storm_stft")
    return spec


def istft(spec, length, peak=None, n_fft=510, hop=128, spec_factor=1.0, spec_abs_exponent=1.0, window="hann", lengths=None):
    """complex64 [B, F, T] -> wav [B, length] = istft(spec_back(spec)) * peak; with `lengths`, row b is
    istft(..., length=lengths[b]) and zero past it."""
    B, F, T = spec.shape
    spec = spec.contiguous()
    win, tw = dft_tables(n_fft, spec.device, window)
    wav = torch.empty((B, length), dtype=torch.float32, device=spec.device)
    frames = torch.empty((B, T, n_fft), dtype=torch.float32, device=spec.device)
    rl = _row_len(lengths, spec)
    L.check(L.lib().storm_istft(L.ptr(_r(spec)), L.ptr(peak), L.ptr(wav), L.ptr(frames), L.ptr(win), L.ptr(tw), B, T,
                                length, wav.stride(0), n_fft, hop, float(spec_factor), float(spec_abs_exponent), L.ptr(rl),
                                L.stream()), "storm_istft")
    return wav
