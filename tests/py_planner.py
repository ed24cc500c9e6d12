"""TEST INFRASTRUCTURE: independent Python restatement of the C++ planner (storm_amd/csrc/ncsnpp_graph.hip): parameter arena
layout + fused op program + liveness-based workspace allocator for one NCSN++ forward (ncsnpp.py:281-450).
tests/test_net.py compares the op list the C ABI plans (storm_ncsnpp_program) with this one, bit for bit."""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from storm_amd import _lib as L
from storm_amd.backbones.plan import (ALIGN, BUF_IN0, BUF_OUT, BUF_PARAMS, BUF_T, BUF_WS, N_BUFS, OP_ATTENTION, OP_CONV, OP_DENSE,
                                      OP_FIR_DOWN, OP_FIR_UP, OP_GN_APPLY, OP_GN_FINALIZE, OP_GN_STATS, OP_MEMSET, OP_OUTPUT_HEAD,
                                      OP_PACK_INPUT, OP_SOFTMAX, OP_TEMB, OP_INPUT_PYRAMID, OP_OUTPUT_PYRAMID, NCSNppConfig, _up, module_list)


# ------------------------------------------------------------------------------------------
# Parameter arena
# ------------------------------------------------------------------------------------------
@dataclass
class ParamEntry:
    key: str                 # arena key
    kind: str                # conv | nin | f32 | f32sum | dense_w | dense_b
    sources: Tuple[str, ...]  # state_dict names
    shape: Tuple[int, ...]   # packed shape
    offset: int = 0
    nbytes: int = 0


class ParamLayout:
    """Byte layout of the packed parameter arena for (cfg, dtype)."""

    def __init__(self, cfg: NCSNppConfig, dtype_code: int):
        self.cfg, self.dtype = cfg, dtype_code
        self.esize = 2 if dtype_code == L.BF16 else 4
        self.per16 = 16 // self.esize
        self.entries: Dict[str, ParamEntry] = {}
        self.size = 0
        self.dense_rows = 0          # total rows of the concatenated Dense_0 matrix
        self.dense_off: Dict[int, int] = {}   # module idx -> row offset
        self._build()

    def _add(self, key, kind, sources, shape, esize):
        n = esize
        for s in shape:
            n *= s
        e = ParamEntry(key, kind, tuple(sources), tuple(shape), self.size, n)
        self.entries[key] = e
        self.size = _up(self.size + n, ALIGN)
        return e

    def conv(self, name, Cout, Cin, taps):
        CoutP, CinP = _up(Cout, 32), _up(Cin, 2 * self.per16)
        return self._add(name, "conv", [name], (taps, CoutP, CinP), self.esize)

    def nin(self, name, Cc):
        return self._add(name, "nin", [name], (1, _up(Cc, 32), Cc), self.esize)

    def f32(self, name, shape):
        return self._add(name, "f32", [name], shape, 4)

    def _build(self):
        cfg = self.cfg
        total = cfg.total_channels
        self.f32("output_layer.weight", (2, total))
        self.f32("output_layer.bias", (2,))
        dense_srcs = []
        for idx, (kind, p) in enumerate(module_list(cfg)):
            k = f"all_modules.{idx}."
            if kind == "gfp":
                self.f32(k + "W", (p["n"],))
            elif kind == "linear":
                self.f32(k + "weight", (p["o"], p["i"])); self.f32(k + "bias", (p["o"],))
            elif kind == "conv3":
                self.conv(k + "weight", p["o"], p["i"], 9); self.f32(k + "bias", (p["o"],))
            elif kind == "gn":
                self.f32(k + "weight", (p["c"],)); self.f32(k + "bias", (p["c"],))
            elif kind == "combine":
                self.conv(k + "Conv_0.weight", p["o"], p["i"], 1); self.f32(k + "Conv_0.bias", (p["o"],))
            elif kind == "attn":
                c = p["c"]
                self.f32(k + "GroupNorm_0.weight", (c,)); self.f32(k + "GroupNorm_0.bias", (c,))
                for j in range(4):
                    self.nin(k + f"NIN_{j}.W", c); self.f32(k + f"NIN_{j}.b", (c,))
            elif kind == "res":
                i, o = p["i"], p["o"]
                self.f32(k + "GroupNorm_0.weight", (i,)); self.f32(k + "GroupNorm_0.bias", (i,))
                self.conv(k + "Conv_0.weight", o, i, 9); self.f32(k + "Conv_0.bias", (o,))
                self.f32(k + "GroupNorm_1.weight", (o,)); self.f32(k + "GroupNorm_1.bias", (o,))
                self.conv(k + "Conv_1.weight", o, o, 9)
                if i != o or p["resample"]:
                    self.conv(k + "Conv_2.weight", o, i, 1)
                    self._add(k + "bias12", "f32sum", [k + "Conv_1.bias", k + "Conv_2.bias"], (o,), 4)
                else:
                    self.f32(k + "Conv_1.bias", (o,))
                if cfg.conditional:
                    self.dense_off[idx] = self.dense_rows
                    self.dense_rows += o
                    dense_srcs.append(k + "Dense_0")
        if cfg.conditional:
            self._add("dense.weight", "dense_w", [s + ".weight" for s in dense_srcs], (self.dense_rows, 4 * cfg.nf), 4)
            self._add("dense.bias", "dense_b", [s + ".bias" for s in dense_srcs], (self.dense_rows,), 4)

    def off(self, key):
        return self.entries[key].offset


# ------------------------------------------------------------------------------------------
# Program builder
# ------------------------------------------------------------------------------------------
class _Arena:
    """First-fit offset allocator with coalescing free list (activations are reused aggressively:
    a 4-s batch-16 forward peaks at a few GB instead of the ~60 GB a bump allocator would take)."""

    def __init__(self):
        self.free: List[List[int]] = []     # [off, size]
        self.top = 0
        self.live: Dict[int, int] = {}

    def alloc(self, nbytes):
        n = _up(max(nbytes, 1), ALIGN)
        for k, (off, size) in enumerate(self.free):
            if size >= n:
                if size == n:
                    self.free.pop(k)
                else:
                    self.free[k] = [off + n, size - n]
                self.live[off] = n
                return off
        # extend: merge with a trailing free block if it touches the top
        if self.free and self.free[-1][0] + self.free[-1][1] == self.top:
            off, size = self.free.pop()
            self.top = off + n
        else:
            off = self.top
            self.top += n
        self.live[off] = n
        return off

    def release(self, off):
        n = self.live.pop(off)
        self.free.append([off, n])
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1][1] += s
            else:
                merged.append([o, s])
        self.free = merged


@dataclass
class Act:
    """An NHWC activation living in the workspace."""
    off: int
    H: int
    W: int
    C: int
    external: Optional[int] = None     # buffer slot if not in the workspace
    part: Optional[int] = None         # fused GroupNorm partials [B][tiles][C][2] fp32 left by the producing conv
    tiles: int = 0


class Program:
    def __init__(self, cfg: NCSNppConfig, layout: ParamLayout, B: int, F: int, T: int, fuse_stats: bool = True,
                 fuse_apply: bool = True, fused_attention: bool = True):
        self.cfg, self.layout, self.B, self.F, self.T = cfg, layout, B, F, T
        self.fused_attention = fused_attention
        self.fuse_stats = fuse_stats       # GroupNorm statistics from the producing conv's epilogue (no stats pass)
        self.fuse_apply = fuse_apply and fuse_stats   # GroupNorm apply + SiLU inside the consuming conv's operand load
        self.dtype = layout.dtype
        self.esize = layout.esize
        self.ops: List[L.Op] = []
        self.arena = _Arena()
        self.n_in = cfg.total_channels // 2
        nlev = len(cfg.ch_mult)
        if F % (1 << (nlev - 1)) or T % (1 << (nlev - 1)):
            raise ValueError(f"spectrogram {F}x{T} must be divisible by {1 << (nlev - 1)}")
        self.flops = 0
        self._build()
        self.ws_bytes = self.arena.top
        self.op_array = (L.Op * len(self.ops))(*self.ops)

    # ---- helpers -------------------------------------------------------------------------
    def _op(self, code):
        op = L.Op()
        op.code = code
        for j in range(L.OP_NPTR):
            op.p[j].buf = -1
        self.ops.append(op)
        return op

    @staticmethod
    def _ref(op, j, buf, off):
        op.p[j].buf, op.p[j].off = buf, off

    def _ws(self, op, j, act_or_off):
        off = act_or_off.off if isinstance(act_or_off, Act) else act_or_off
        self._ref(op, j, BUF_WS, off)

    def _par(self, op, j, key):
        self._ref(op, j, BUF_PARAMS, self.layout.off(key))

    def new_act(self, H, W, Cc, esize=None):
        n = self.B * H * W * Cc * (esize or self.esize)
        return Act(self.arena.alloc(n), H, W, Cc)

    def free(self, a: Act):
        self.arena.release(a.off)
        if a.part is not None:
            self.arena.release(a.part)

    def new_stats(self, G):
        off = self.stats_cursor
        self.stats_cursor += _up(self.B * G * 2 * 8, ALIGN)
        assert self.stats_cursor <= self.stats_off + self.stats_bytes
        return off

    # ---- op emitters -----------------------------------------------------------------------
    def gn(self, xa: Act, xb: Optional[Act], wkey, bkey, silu=True, resample=0):
        Cc = xa.C + (xb.C if xb else 0)
        G = min(Cc // 4, 32)
        st = self.new_stats(G)
        if xa.part is not None and (xb is None or xb.part is not None):
            op = self._op(OP_GN_FINALIZE)          # statistics were accumulated by the producing conv epilogues
            self._ws(op, 0, xa.part)
            if xb:
                self._ws(op, 1, xb.part)
            self._ws(op, 2, st)
            op.i[0], op.i[1], op.i[2], op.i[3] = xa.C, xa.tiles, (xb.C if xb else 0), (xb.tiles if xb else 0)
            op.i[4], op.i[5] = self.B, G
        else:
            op = self._op(OP_GN_STATS)
            self._ws(op, 0, xa)
            if xb:
                self._ws(op, 1, xb)
            self._ws(op, 2, st)
            op.i[0], op.i[1], op.i[2], op.i[3], op.i[4] = xa.C, (xb.C if xb else 0), self.B, xa.H * xa.W, G
        OH, OW = (2 * xa.H, 2 * xa.W) if resample == 1 else ((xa.H // 2, xa.W // 2) if resample == 2 else (xa.H, xa.W))
        out = self.new_act(OH, OW, Cc)
        raw = self.new_act(OH, OW, Cc) if resample else None
        op = self._op(OP_GN_APPLY)
        self._ws(op, 0, xa)
        if xb:
            self._ws(op, 1, xb)
        self._ws(op, 2, st)
        self._par(op, 3, wkey)
        self._par(op, 4, bkey)
        self._ws(op, 5, out)
        if raw:
            self._ws(op, 6, raw)
        op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5] = xa.C, (xb.C if xb else 0), self.B, xa.H, xa.W, G
        op.i[6], op.i[7] = int(silu), resample
        op.f[0] = 1e-6
        return out, raw

    def gn_affine(self, xa: Act, xb: Optional[Act], wkey, bkey):
        """GroupNorm as a per-(batch, channel) affine (scale, shift) table for a conv that fuses the apply
        (+SiLU) into its operand load: one tiny finalize launch, no pass over the activation."""
        assert xa.part is not None and (xb is None or xb.part is not None)
        Cc = xa.C + (xb.C if xb else 0)
        G = min(Cc // 4, 32)
        st = self.new_stats(G)
        ss = self.arena.alloc(self.B * Cc * 2 * 4)
        op = self._op(OP_GN_FINALIZE)
        self._ws(op, 0, xa.part)
        if xb:
            self._ws(op, 1, xb.part)
        self._ws(op, 2, st)
        self._par(op, 3, wkey)
        self._par(op, 4, bkey)
        self._ws(op, 5, ss)
        op.i[0], op.i[1], op.i[2], op.i[3] = xa.C, xa.tiles, (xb.C if xb else 0), (xb.tiles if xb else 0)
        op.i[4], op.i[5], op.i[6] = self.B, G, xa.H * xa.W
        op.f[0] = 1e-6
        return ss

    def conv(self, segs, Cout, H, W, outC=None, bias_key=None, tbias=None, skip: Optional[Act] = None, scale=1.0,
             out_f32=False, out_bstride=-1, src0_bstride=-1, want_part=False):
        """segs: list of dict(a=Act|(buf,off,C), b=Act|None, w=('par',key)|('ws',off), CinP, rows, taps,
        w_bstride, w_tapstride)."""
        outC = outC or _up(Cout, 8)
        out = self.new_act(H, W, outC, 4 if out_f32 else None)
        op = self._op(OP_CONV)
        op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5] = len(segs), self.B, H, W, outC, Cout
        op.i[7] = int(out_f32)
        for g, s in enumerate(segs):
            a = s["a"]
            if isinstance(a, Act):
                self._ws(op, 3 * g, a)
                Ca = a.C
            else:
                self._ref(op, 3 * g, a[0], a[1])
                Ca = a[2]
            Cb = 0
            if s.get("b") is not None:
                self._ws(op, 3 * g + 1, s["b"])
                Cb = s["b"].C
            kindw, wv = s["w"]
            if kindw == "par":
                self._par(op, 3 * g + 2, wv)
            else:
                self._ws(op, 3 * g + 2, wv)
            q = 8 + 7 * g
            op.i[q], op.i[q + 1], op.i[q + 2], op.i[q + 3], op.i[q + 4] = Ca, Cb, s["CinP"], s["rows"], s["taps"]
            op.i[q + 5], op.i[q + 6] = s.get("w_bstride", 0), s.get("w_tapstride", s["CinP"] * s["rows"])
            self.flops += 2 * self.B * H * W * Cout * (Ca + Cb) * s["taps"]
        op.i[22], op.i[23] = src0_bstride, out_bstride
        if segs[0].get("gn") is not None:
            self._ws(op, 11, segs[0]["gn"])
            op.f[1] = 1.0 if segs[0].get("gn_silu", True) else 0.0
        self._ws(op, 6, out)
        if bias_key:
            self._par(op, 7, bias_key)
        if tbias is not None:
            self._ref(op, 8, BUF_WS, tbias[0])
            op.i[6] = tbias[1]
        if skip is not None:
            assert skip.C == outC and skip.H == H and skip.W == W
            self._ws(op, 9, skip)
        op.f[0] = scale
        if want_part:
            any9 = any(sg["taps"] == 9 for sg in segs)
            out.tiles = (-(-W // 32)) * (-(-H // 8)) if any9 else -(-(H * W) // 256)
            out.part = self.arena.alloc(self.B * out.tiles * outC * 2 * 4)
            self._ws(op, 10, out.part)
        nb = self.splitk_bytes(segs, H, W, outC, out_f32)
        if nb > 0:                                  # scratch of the split-K launch: live for this op only
            off = self.arena.alloc(nb)
            self._ws(op, 12, off)
            self.arena.release(off)
            op.f[2] = float(nb // (self.B * H * W * outC * 4))
        return out

    def splitk_bytes(self, segs, H, W, outC, out_f32):
        """conv_pipe.hip: conv_splitk_slices / conv_splitk_bytes restated - a 16-bit 3x3 layer with > 128 output channels and at most
        8 128-cout tiles PER IMAGE - or at most 64 in the whole launch (small calls) - splits its nine-tap chunks (64 channels each) into 4 (or 2) slices, one fp32 slab [B][H][W][outC]
        per slice."""
        if self.esize != 2 or out_f32 or outC <= 128 or segs[0]["taps"] != 9:
            return 0
        # what the pipelined kernel covers (build_pipe_params): a nine-tap segment first, then at most one one-tap segment without
        # a fused GroupNorm, shared weights, <= 4 weight runs, <= 36 chunks, images and weight matrices below 2 GiB
        runs = chunks = 0
        for g, s in enumerate(segs):
            if s["taps"] not in (9, 1) or s.get("w_bstride", 0) != 0 or (g > 0 and s["taps"] == 9):
                return 0
            if s["taps"] == 1 and s.get("gn") is not None:
                return 0
            a = s["a"]
            Ca = a.C if isinstance(a, Act) else a[2]
            Cb = s["b"].C if s.get("b") is not None else 0
            tapstride = s.get("w_tapstride", s["CinP"] * s["rows"])
            for part, Cc in enumerate((Ca, Cb)):
                if part == 1 and Cb == 0:
                    break
                runs += 1
                chunks += -(-Cc // 64)
                wc0 = 0 if part == 0 else Ca
                if H * W * Cc * 2 >= 2 ** 31 or (s["taps"] * tapstride - wc0) * 2 >= 2 ** 31:
                    return 0
        if runs > 4 or chunks > 36:
            return 0
        a0 = segs[0]["a"]
        Ca0 = a0.C if isinstance(a0, Act) else a0[2]
        Cb0 = segs[0]["b"].C if segs[0].get("b") is not None else 0
        per_image = (-(-W // 32)) * (-(-H // 8)) * (-(-outC // 128))
        if per_image > 8 and per_image * self.B > 64:                 # the image rule (batch-invariant) or the small-call rule (<= 64 workgroups)
            return 0
        n9 = -(-Ca0 // 64) + (-(-Cb0 // 64) if Cb0 else 0)
        S = 4 if n9 >= 4 else 2 if n9 >= 2 else 0
        return S * self.B * H * W * outC * 4 if S >= 2 else 0

    def wseg(self, a, key, taps, b=None, gn=None):
        e = self.layout.entries[key]
        return dict(a=a, b=b, w=("par", key), CinP=e.shape[2], rows=e.shape[1], taps=taps, gn=gn)

    # ---- blocks ----------------------------------------------------------------------------
    def resblock(self, idx, p, xa: Act, xb: Optional[Act] = None, resample=0):
        """ResnetBlockBigGANpp.forward (layerspp.py:242-274) as 6-7 fused ops."""
        k = f"all_modules.{idx}."
        o = p["o"]
        tb = None
        if self.cfg.conditional:
            tb = (self.dense_out + 4 * self.layout.dense_off[idx], self.layout.dense_rows)
        inv = 1.0 / math.sqrt(2.0)
        fuse = self.fuse_apply and xa.part is not None and (xb is None or xb.part is not None)
        xr = None
        if fuse and not resample:
            # GroupNorm_0 + SiLU ride in Conv_0's operand load (no normalised copy of x in HBM)
            ss0 = self.gn_affine(xa, xb, k + "GroupNorm_0.weight", k + "GroupNorm_0.bias")
            u = self.conv([self.wseg(xa, k + "Conv_0.weight", 9, b=xb, gn=ss0)], o, xa.H, xa.W, bias_key=k + "Conv_0.bias",
                          tbias=tb, want_part=True)
            self.arena.release(ss0)
        else:
            a, xr = self.gn(xa, xb, k + "GroupNorm_0.weight", k + "GroupNorm_0.bias", True, resample)
            u = self.conv([self.wseg(a, k + "Conv_0.weight", 9)], o, a.H, a.W, bias_key=k + "Conv_0.bias", tbias=tb,
                          want_part=self.fuse_stats)
            self.free(a)
        if fuse:
            ss1 = self.gn_affine(u, None, k + "GroupNorm_1.weight", k + "GroupNorm_1.bias")
            s1 = self.wseg(u, k + "Conv_1.weight", 9, gn=ss1)
            a2 = None
        else:
            a2, _ = self.gn(u, None, k + "GroupNorm_1.weight", k + "GroupNorm_1.bias", True, 0)
            self.free(u)
            s1 = self.wseg(a2, k + "Conv_1.weight", 9)
        H2, W2 = u.H, u.W
        if (k + "Conv_2.weight") in self.layout.entries:
            if xr is not None:
                s2 = self.wseg(xr, k + "Conv_2.weight", 1)
            else:
                s2 = self.wseg(xa, k + "Conv_2.weight", 1, b=xb)
            out = self.conv([s1, s2], o, H2, W2, bias_key=k + "bias12", scale=inv, want_part=self.fuse_stats)
        else:
            assert xb is None and xr is None and xa.C == o
            out = self.conv([s1], o, H2, W2, bias_key=k + "Conv_1.bias", skip=xa, scale=inv, want_part=self.fuse_stats)
        if fuse:
            self.arena.release(ss1)
            self.free(u)
        else:
            self.free(a2)
        if xr is not None:
            self.free(xr)
        return out

    def attn_scratch_bytes(self, Lp, Cc):
        """attention.hip: attn_splits / attn_scratch_bytes restated - a 16-bit call with fewer than 128 query blocks (128 queries each)
        splits its key tiles (32 keys) into 2 / 4 / 8 ranges while the split launch fits 256 workgroups and a range keeps >= 4 tiles;
        scratch = fp32 partial outputs [S][B][L][C] + (maximum, sum) [S][B][L][2]"""
        if self.esize != 2:
            return 0
        ntiles, wgs = -(-Lp // 32), -(-Lp // 128) * self.B
        cus = getattr(self, "cus", 256)                    # device_cus(): 256 on MI355X and in the simulator
        if wgs >= cus // 2:
            return 0
        S = 1
        while S < 8 and wgs * S * 2 <= cus and ntiles // (S * 2) >= 4:
            S *= 2
        return S * self.B * Lp * (Cc + 2) * 4 if S >= 2 else 0

    def attnblock(self, idx, p, x: Act):
        """AttnBlockpp.forward (layerspp.py:75-91): GN -> q,k,v (NIN) -> softmax(q k^T / sqrt C) v -> NIN_3 -> skip."""
        k = f"all_modules.{idx}."
        Cc, Lp = p["c"], x.H * x.W
        Lp8 = _up(Lp, 8)                 # row padding of the [L][L] score / probability / v^T matrices
        h, _ = self.gn(x, None, k + "GroupNorm_0.weight", k + "GroupNorm_0.bias", silu=False)
        hl = Act(h.off, 1, Lp, Cc)
        q = self.conv([self.wseg(hl, k + "NIN_0.W", 1)], Cc, 1, Lp, bias_key=k + "NIN_0.b")
        kk = self.conv([self.wseg(hl, k + "NIN_1.W", 1)], Cc, 1, Lp, bias_key=k + "NIN_1.b")
        # v^T[c][j] = sum_c' Wv^T[c][c'] h[j][c']: the packed NIN_2 matrix is the "pixel" operand, h the weights.
        e2 = self.layout.entries[k + "NIN_2.W"]
        vT = self.conv([dict(a=(BUF_PARAMS, e2.offset, Cc), w=("ws", h.off), CinP=Cc, rows=Lp, taps=1,
                             w_bstride=Lp * Cc)], Lp, 1, Cc, outC=Lp8, src0_bstride=0)
        self.free(h)
        if self.fused_attention and L.lib().storm_attention_supported(Cc, self.dtype):
            # flash-style kernel: softmax(q k^T / sqrt C) v + b_v in one launch, the [L][L] scores never reach HBM
            o = self.new_act(1, Lp, Cc)
            op = self._op(OP_ATTENTION)
            self._ws(op, 0, q); self._ws(op, 1, kk); self._ws(op, 2, vT); self._par(op, 3, k + "NIN_2.b"); self._ws(op, 4, o)
            op.i[0], op.i[1], op.i[2], op.i[3] = self.B, Lp, Cc, Lp8
            op.f[0] = float(int(Cc) ** (-0.5))
            nb = self.attn_scratch_bytes(Lp, Cc)        # key-range split of small calls: scratch live for this op only
            if nb > 0:
                off = self.arena.alloc(nb)
                self._ws(op, 5, off)
                op.i[4] = nb
                self.arena.release(off)
            self.flops += 4 * self.B * Lp * Lp * Cc
            self.free(q); self.free(kk); self.free(vT)
        else:
            S = self.conv([dict(a=q, w=("ws", kk.off), CinP=Cc, rows=Lp, taps=1, w_bstride=Lp * Cc)], Lp, 1, Lp, outC=Lp8,
                          scale=float(int(Cc) ** (-0.5)), out_f32=True)
            self.free(q); self.free(kk)
            P = self.new_act(1, Lp, Lp8)
            op = self._op(OP_SOFTMAX)
            self._ws(op, 0, S); self._ws(op, 1, P)
            op.i[0], op.i[1], op.i[2] = self.B * Lp, Lp, Lp8
            self.free(S)
            # h = P v (+ b_v: rows of P sum to one, so the NIN_2 bias passes through unchanged)
            o = self.conv([dict(a=P, w=("ws", vT.off), CinP=Lp8, rows=Cc, taps=1, w_bstride=Cc * Lp8)], Cc, 1, Lp,
                          bias_key=k + "NIN_2.b")
            self.free(P); self.free(vT)
        xl = Act(x.off, 1, Lp, Cc)
        out = self.conv([self.wseg(o, k + "NIN_3.W", 1)], Cc, 1, Lp, bias_key=k + "NIN_3.b", skip=xl,
                        scale=1.0 / math.sqrt(2.0), want_part=self.fuse_stats)
        self.free(o)
        return Act(out.off, x.H, x.W, Cc, part=out.part, tiles=out.tiles)

    # ---- whole network ---------------------------------------------------------------------
    def _build(self):
        cfg, B, F, T = self.cfg, self.B, self.F, self.T
        mods = module_list(cfg)
        nres, total = len(cfg.ch_mult), cfg.total_channels
        # statistics arena: every GroupNorm gets its own [B][G][2] fp64 slot, zeroed by one memset
        n_gn = sum({"res": 2, "attn": 1, "gn": 1}.get(kind, 0) for kind, _ in mods)
        self.stats_bytes = n_gn * _up(B * 32 * 2 * 8, ALIGN)
        self.stats_off = self.arena.alloc(self.stats_bytes)
        self.stats_cursor = self.stats_off
        op = self._op(OP_MEMSET)
        self._ws(op, 0, self.stats_off)
        op.i[0] = self.stats_bytes

        # the input pyramid is a function of the network input alone: packing + every FIR x2 down step ahead of the U-Net, up to three steps per launch
        ips = [self.new_act(F >> lvl, T >> lvl, 8) for lvl in range(nres)]
        l0 = 0
        while l0 == 0 or l0 < nres - 1:
            nl = min(4, nres - l0)
            op = self._op(OP_INPUT_PYRAMID)
            if l0 == 0:
                for j in range(self.n_in):
                    self._ref(op, j, BUF_IN0 + j, 0)
            for k in range(nl):
                self._ws(op, 3 + k, ips[l0 + k])
            op.i[0], op.i[1], op.i[2], op.i[3], op.i[4] = (self.n_in if l0 == 0 else 0), B, F >> l0, T >> l0, nl
            l0 += 3
        x0 = ips[0]

        midx = 1
        if cfg.conditional:
            temb = self.arena.alloc(B * 4 * cfg.nf * 4)
            op = self._op(OP_TEMB)
            self._ref(op, 0, BUF_T, 0)
            self._par(op, 1, "all_modules.0.W")
            self._par(op, 2, "all_modules.1.weight"); self._par(op, 3, "all_modules.1.bias")
            self._par(op, 4, "all_modules.2.weight"); self._par(op, 5, "all_modules.2.bias")
            self._ws(op, 6, temb)
            op.i[0], op.i[1] = B, cfg.nf
            self.dense_out = self.arena.alloc(B * self.layout.dense_rows * 4)
            op = self._op(OP_DENSE)
            self._ws(op, 0, temb); self._par(op, 1, "dense.weight"); self._par(op, 2, "dense.bias")
            self._ws(op, 3, self.dense_out)
            op.i[0], op.i[1], op.i[2] = B, self.layout.dense_rows, 4 * cfg.nf
            midx = 3

        ip = x0
        k = f"all_modules.{midx}."
        hs = [self.conv([self.wseg(x0, k + "weight", 9)], cfg.nf, F, T, bias_key=k + "bias", want_part=self.fuse_stats)]
        midx += 1
        for lvl in range(nres):
            for _ in range(cfg.num_res_blocks):
                h = self.resblock(midx, mods[midx][1], hs[-1]); midx += 1
                if h.H in cfg.attn_resolutions:                     # ncsnpp.py:338 (frequency axis)
                    h2 = self.attnblock(midx, mods[midx][1], h); midx += 1
                    self.free(h); h = h2
                hs.append(h)
            if lvl != nres - 1:
                h = self.resblock(midx, mods[midx][1], hs[-1], resample=2); midx += 1
                self.free(ip); ip = ips[lvl + 1]
                kk = f"all_modules.{midx}."                          # Combine (layerspp.py:52-57), method 'sum'
                hc = self.conv([self.wseg(ip, kk + "Conv_0.weight", 1)], h.C, h.H, h.W, bias_key=kk + "Conv_0.bias", skip=h,
                               want_part=self.fuse_stats)
                midx += 1
                self.free(h)
                hs.append(hc)
        self.free(ip)
        h = hs[-1]
        h1 = self.resblock(midx, mods[midx][1], h); midx += 1
        h2 = self.attnblock(midx, mods[midx][1], h1); midx += 1
        self.free(h1)
        h = self.resblock(midx, mods[midx][1], h2); midx += 1
        self.free(h2)
        phs = []
        for lvl in reversed(range(nres)):
            for _ in range(cfg.num_res_blocks + 1):
                skip = hs.pop()
                hn = self.resblock(midx, mods[midx][1], h, skip); midx += 1
                self.free(h); self.free(skip)
                h = hn
            if h.H in cfg.attn_resolutions:                          # ncsnpp.py:385
                hn = self.attnblock(midx, mods[midx][1], h); midx += 1
                self.free(h); h = hn
            kg, kc = f"all_modules.{midx}.", f"all_modules.{midx + 1}."
            if self.fuse_apply and h.part is not None:
                ssp = self.gn_affine(h, None, kg + "weight", kg + "bias")
                ph = self.conv([self.wseg(h, kc + "weight", 9, gn=ssp)], total, h.H, h.W, outC=8, bias_key=kc + "bias")
                self.arena.release(ssp)
            else:
                a, _ = self.gn(h, None, kg + "weight", kg + "bias", True, 0)
                ph = self.conv([self.wseg(a, kc + "weight", 9)], total, h.H, h.W, outC=8, bias_key=kc + "bias")
                self.free(a)
            midx += 2
            phs.append(ph)                                           # (coarsest first; the up chain and the head are one launch at the end)
            if lvl != 0:
                hn = self.resblock(midx, mods[midx][1], h, resample=1); midx += 1
                self.free(h); h = hn
        assert not hs and midx == len(mods)
        self.free(h)
        op = self._op(OP_OUTPUT_PYRAMID)
        for k in range(nres):
            self._ws(op, k, phs[nres - 1 - k])
        if cfg.conditional:
            self._ref(op, 8, BUF_T, 0)
        self._par(op, 9, "output_layer.weight"); self._par(op, 10, "output_layer.bias")
        self._ref(op, 11, BUF_OUT, 0)
        op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5] = total, B, F, T, 0, nres
        for a in phs:
            self.free(a)
