"""NCSN++ score network on the HIP engine.

Drop-in for the reference backbone plugin (sgmse/backbones/ncsnpp.py:36-450): same registry
names, constructor keywords, ``state_dict`` key names/shapes (``all_modules.<i>.<Layer>.<param>``,
``output_layer.*``) and ``forward(x: complex64[B,C,F,T], time_cond: float32[B]) ->
complex64[B,1,F,T]`` contract.  The modules below only HOLD parameters; ``forward`` plans the
whole network once per (B, F, T, dtype) (plan.py) and runs it with one ``storm_program_run``
call.  There is no PyTorch/CPU execution path.
"""
import math
import os
import threading

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .plan import NCSNppConfig, module_list
from .shared import BackboneRegistry
import ctypes as C


def _default_init(shape, scale=1.0):
    """DDPM / JAX variance_scaling(scale, 'fan_avg', 'uniform') (layers.py:52-91)."""
    scale = 1e-10 if scale == 0 else scale
    receptive = np.prod(shape) / shape[0] / shape[1]
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    variance = scale / ((fan_in + fan_out) / 2)
    return (torch.rand(*shape) * 2.0 - 1.0) * np.sqrt(3 * variance)


def _conv(i, o, k, init_scale=1.0):
    c = nn.Conv2d(i, o, k, padding=k // 2)
    c.weight.data = _default_init(c.weight.shape, init_scale)
    nn.init.zeros_(c.bias)
    return c


def _dense(i, o):
    d = nn.Linear(i, o)
    d.weight.data = _default_init(d.weight.shape)
    nn.init.zeros_(d.bias)
    return d


def _gn(c):
    return nn.GroupNorm(num_groups=min(c // 4, 32), num_channels=c, eps=1e-6)


class _Holder(nn.Module):
    """Parameter container; the engine reads the parameters, nothing is executed here."""

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: run the enclosing NCSNpp, not its sub-modules")


class GaussianFourierProjection(_Holder):
    def __init__(self, embedding_size, scale):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class NIN(_Holder):
    def __init__(self, c, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(_default_init((c, c), init_scale))
        self.b = nn.Parameter(torch.zeros(c))


class AttnBlockpp(_Holder):
    def __init__(self, c, init_scale=0.0):
        super().__init__()
        self.GroupNorm_0 = _gn(c)
        self.NIN_0, self.NIN_1, self.NIN_2 = NIN(c), NIN(c), NIN(c)
        self.NIN_3 = NIN(c, init_scale)


class ResnetBlockBigGANpp(_Holder):
    def __init__(self, i, o, temb_dim, resample, init_scale=0.0):
        super().__init__()
        self.GroupNorm_0 = _gn(i)
        self.Conv_0 = _conv(i, o, 3)
        self.Dense_0 = _dense(temb_dim, o)
        self.GroupNorm_1 = _gn(o)
        self.Conv_1 = _conv(o, o, 3, init_scale)
        if i != o or resample:
            self.Conv_2 = _conv(i, o, 1)


class Combine(_Holder):
    def __init__(self, i, o):
        super().__init__()
        self.Conv_0 = _conv(i, o, 1)


@BackboneRegistry.register("ncsnpp")
class NCSNpp(nn.Module):
    """NCSN++ (27.8 M parameters with the defaults)."""

    def __init__(self, scale_by_sigma=True, nonlinearity="swish", nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1,
                 attn_resolutions=(0,), resamp_with_conv=True, conditional=True, fir=True, fir_kernel=(1, 3, 3, 1),
                 skip_rescale=True, resblock_type="biggan", progressive="output_skip", progressive_input="input_skip",
                 progressive_combine="sum", init_scale=0.0, fourier_scale=16, image_size=256, embedding_type="fourier",
                 input_channels=4, spatial_channels=1, dropout=0.0, centered=False, discriminative=False, **kwargs):
        super().__init__()
        unsupported = []
        if nonlinearity != "swish": unsupported.append(f"nonlinearity={nonlinearity}")
        if not fir or list(fir_kernel) != [1, 3, 3, 1]: unsupported.append("fir / fir_kernel")
        if not skip_rescale: unsupported.append("skip_rescale=False")
        if resblock_type.lower() != "biggan": unsupported.append(f"resblock_type={resblock_type}")
        if progressive.lower() != "output_skip": unsupported.append(f"progressive={progressive}")
        if progressive_input.lower() != "input_skip": unsupported.append(f"progressive_input={progressive_input}")
        if progressive_combine.lower() != "sum": unsupported.append(f"progressive_combine={progressive_combine}")
        if embedding_type.lower() != "fourier": unsupported.append(f"embedding_type={embedding_type}")
        if spatial_channels != 1: unsupported.append(f"spatial_channels={spatial_channels}")
        if dropout != 0.0: unsupported.append("dropout (inference engine)")
        if centered: unsupported.append("centered=True")
        if not discriminative and (not conditional or not scale_by_sigma):
            unsupported.append("conditional / scale_by_sigma = False on a score network")
        if unsupported:
            raise NotImplementedError("storm_amd NCSNpp covers the StoRM hot-path configuration only; unsupported: "
                                      + ", ".join(unsupported))
        self.FORCE_STFT_OUT = False
        if discriminative:
            print("Running NCSN++ as discriminative backbone")
            input_channels = 2
        if nf % 8:
            raise NotImplementedError("nf must be a multiple of 8 (NHWC channel octets)")
        self.cfg = NCSNppConfig(nf=nf, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks,
                                attn_resolutions=tuple(attn_resolutions), image_size=image_size,
                                input_channels=input_channels, discriminative=discriminative,
                                fourier_scale=float(fourier_scale))
        self.nf, self.discriminative, self.input_channels = nf, discriminative, input_channels
        self.spatial_channels = 1
        total = self.cfg.total_channels
        self.output_layer = nn.Conv2d(total, 2, 1)
        mods = []
        for kind, p in module_list(self.cfg):
            if kind == "gfp":
                mods.append(GaussianFourierProjection(p["n"], fourier_scale))
            elif kind == "linear":
                mods.append(_dense(p["i"], p["o"]))
            elif kind == "conv3":
                first = p["i"] == total and p["o"] == nf and len(mods) <= 3
                mods.append(_conv(p["i"], p["o"], 3, 1.0 if first else init_scale))
            elif kind == "res":
                mods.append(ResnetBlockBigGANpp(p["i"], p["o"], 4 * nf, p["resample"], init_scale))
            elif kind == "combine":
                mods.append(Combine(p["i"], p["o"]))
            elif kind == "attn":
                mods.append(AttnBlockpp(p["c"], init_scale))
            elif kind == "gn":
                mods.append(_gn(p["c"]))
        self.all_modules = nn.ModuleList(mods)
        self.compute_dtype = torch.float32
        self.negate_output = False        # ScoreModel folds the "score = -dnn(...)" sign into the output head
        self._handles = {}                # dtype code -> (engine handle, arena tensor, params version, device)
        self._workspaces = {}             # (device, dtype, stream) -> ONE grow-only scratch tensor (the workspace holds no state between calls)
        self._lock = threading.RLock()    # handle / scratch bookkeeping (host threads driving different streams share the module)
        self._graph_mode = -1             # storm_ncsnpp_set_graph: -1 = the library's rule (eager: measured, profiles/r05a_*), 0 eager, 1 replay
        self._param_version = 0
        self.register_load_state_dict_post_hook(lambda m, keys: m.invalidate())

    @staticmethod
    def add_argparse_args(parser):
        return parser

    # ---- engine state ----------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """torch.float32 (exact fp32 MFMA, parity path) or torch.bfloat16 (bf16 MFMA operands and
        activations, fp32 accumulation / statistics / SDE state)."""
        L.dt(dtype)
        self.compute_dtype = dtype
        return self

    MAX_WORKSPACES = 4                    # streams (per device and dtype) whose scratch is kept; least recently used out

    def set_graph(self, mode):
        """HIP-graph replay of the score evaluations (include/storm_hip.h: storm_ncsnpp_set_graph): "auto" / -1 = the library's rule
        (eager launches: the queue never drains on MI355X, profiles/r05a_*), 0 / False = eager launches, 1 / True = replay.  Results are bit-identical either way."""
        self._graph_mode = -1 if mode in ("auto", -1, None) else int(bool(mode))
        for h, _, _, _ in self._handles.values():
            L.check(L.lib().storm_ncsnpp_set_graph(h, self._graph_mode), "storm_ncsnpp_set_graph")
        return self

    def graph_launches(self):
        """hipGraphLaunch calls made so far by this module's engine handles (0 = every evaluation ran as eager launches)"""
        return sum(int(L.lib().storm_ncsnpp_graph_launches(h)) for h, _, _, _ in self._handles.values())

    def group_launches(self):
        """grouped kernel launches made so far by forward_parts_group (0 = every op ran problem by problem)"""
        return sum(int(L.lib().storm_ncsnpp_group_launches(h)) for h, _, _, _ in self._handles.values())

    def invalidate(self):
        """Call after changing parameters in place (e.g. EMA swap): the engine re-packs its weight arena lazily."""
        self._param_version += 1

    def _apply(self, fn, *a, **k):
        self.invalidate()
        self._drop_handles()
        return super()._apply(fn, *a, **k)

    def _drop_handles(self):
        for h, _, _, _ in self._handles.values():
            L.lib().storm_ncsnpp_destroy(h)
        self._handles.clear()
        self._workspaces.clear()

    def __del__(self):
        try:
            self._drop_handles()
        except Exception:
            pass

    def c_config(self):
        """the storm_ncsnpp_config of this network (include/storm_hip.h)"""
        c = L.NcsnppConfig()
        c.nf, c.n_levels, c.num_res_blocks = self.cfg.nf, len(self.cfg.ch_mult), self.cfg.num_res_blocks
        for i, v in enumerate(self.cfg.ch_mult):
            c.ch_mult[i] = v
        c.n_attn = len(self.cfg.attn_resolutions)
        for i, v in enumerate(self.cfg.attn_resolutions):
            c.attn_resolutions[i] = v
        c.image_size, c.input_channels, c.discriminative = self.cfg.image_size, self.cfg.input_channels, int(self.cfg.discriminative)
        return c

    def _get_handle(self, dtype_code, device):
        """The C-ABI network object for (dtype, device): storm_ncsnpp_create over this module's state_dict tensors (the
        planner, the weight packing and the op program all live behind the ABI, csrc/ncsnpp_graph.hip)."""
        ent = self._handles.get(dtype_code)
        if ent is not None and ent[2] == self._param_version and ent[3] == device:
            return ent[0]
        if ent is not None:
            L.lib().storm_ncsnpp_destroy(ent[0])
            self._workspaces.clear()
        cfg = self.c_config()
        lib = L.lib()
        sd = [v.detach().to(device=device, dtype=torch.float32).contiguous() for v in self.state_dict().values()]
        n = lib.storm_ncsnpp_num_tensors(C.byref(cfg))
        if n != len(sd):
            raise L.StormError(f"state_dict has {len(sd)} tensors, the engine's enumeration {n}")
        ptrs = (C.c_void_p * n)(*[L.ptr(t) for t in sd])
        arena = torch.empty(lib.storm_ncsnpp_arena_bytes(C.byref(cfg), dtype_code), dtype=torch.uint8, device=device)
        h = C.c_void_p()
        L.check(lib.storm_ncsnpp_create(C.byref(cfg), ptrs, n, dtype_code, L.ptr(arena), L.stream(), C.byref(h)), "storm_ncsnpp_create")
        L.check(lib.storm_ncsnpp_set_fusion(h, int(os.environ.get("STORM_FUSE_GN_STATS", "1") != "0"),
                                            int(os.environ.get("STORM_FUSE_GN_APPLY", "1") != "0"),
                                            int(os.environ.get("STORM_FUSED_ATTENTION", "1") != "0")), "storm_ncsnpp_set_fusion")
        if not L.is_sim():
            torch.cuda.current_stream().synchronize()    # the fp32 staging copies die with this scope
        L.check(lib.storm_ncsnpp_set_graph(h, self._graph_mode), "storm_ncsnpp_set_graph")
        self._handles[dtype_code] = (h, arena, self._param_version, device)
        return h

    def _get_workspace(self, h, B, F, T, dtype_code, device):
        """Scratch for one forward at (B, F, T).  The workspace carries nothing from one call to the next, so there is ONE
        buffer per (device, dtype, stream), grown to the largest size seen: a ragged stream (17 frame buckets x tail batch sizes)
        holds the memory of its biggest micro-batch, not the sum over shapes; two streams driving one module never share scratch."""
        n = L.lib().storm_ncsnpp_workspace_bytes(h, B, F, T)
        if n < 0:
            raise L.StormError(f"storm_ncsnpp_workspace_bytes: {L.lib().storm_last_error().decode()}")
        key = (str(device), dtype_code, L.stream())
        ws = self._workspaces.pop(key, None)           # (re-inserted below: the dict is kept in least-recently-used order)
        if ws is None or ws.numel() < n:
            ws = None                                  # (release the old buffer before asking the allocator for the larger one)
            # a raw stream handle outlives nothing: scratch of streams that are gone (or whose handle was recycled) must not pile up -
            # keep the few most recently used, drop the rest before growing
            while len(self._workspaces) >= self.MAX_WORKSPACES:
                self._workspaces.pop(next(iter(self._workspaces)))
            ws = torch.empty(n, dtype=torch.uint8, device=device)
        self._workspaces[key] = ws
        return ws

    def workspace_bytes(self, B, F, T, dtype=None, device=None):
        code = L.dt(dtype or self.compute_dtype)
        dev = device or next(self.parameters()).device
        return int(L.lib().storm_ncsnpp_workspace_bytes(self._get_handle(code, dev), B, F, T))

    def program(self, B, F, T, dtype=None):
        """(ops pointer, n_ops, flops) of the planned forward - for profilers (storm_program_run_timed)"""
        code = L.dt(dtype or self.compute_dtype)
        h = self._get_handle(code, next(self.parameters()).device)
        ops, n, fl = C.POINTER(L.Op)(), C.c_int(), C.c_longlong()
        L.check(L.lib().storm_ncsnpp_program(h, B, F, T, C.byref(ops), C.byref(n), C.byref(fl)), "storm_ncsnpp_program")
        return ops, n.value, fl.value

    def release_program(self, ops, dtype=None):
        """unpin an op list obtained from program() (a profiler that sweeps many shapes calls this per shape)"""
        code = L.dt(dtype or self.compute_dtype)
        h = self._get_handle(code, next(self.parameters()).device)
        L.check(L.lib().storm_ncsnpp_release_program(h, ops), "storm_ncsnpp_release_program")

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, x, time_cond=None):
        """x: complex64 [B, input_channels/2, F, T] (x, y[, y_denoised]); time_cond: float32 [B]."""
        if not x.is_complex():
            raise TypeError("NCSNpp expects a complex spectrogram batch [B, C, F, T]")
        B, Cc, F, T = x.shape
        if 2 * Cc != self.cfg.total_channels:
            raise ValueError(f"expected {self.cfg.total_channels // 2} complex input channels, got {Cc}")
        ins = [x[:, c].contiguous() for c in range(Cc)]
        return self.forward_parts(ins, time_cond)

    def forward_parts(self, ins, time_cond=None):
        """Same as forward() but takes the complex channels as separate contiguous [B,F,T] tensors
        (avoids materialising torch.cat([x, y], 1) every score evaluation).  One C-ABI call: storm_ncsnpp_forward."""
        from ..sampling.grouped import grouped_forward_parts
        routed = grouped_forward_parts(self, ins, time_cond)       # (a micro-batch of a grouped stream: evaluated together with the others)
        if routed is not None:
            return routed
        x0 = ins[0]
        B, F, T = x0.shape
        dev = x0.device
        code = L.dt(self.compute_dtype)
        with self._lock:
            h = self._get_handle(code, dev)
            ws = self._get_workspace(h, B, F, T, code, dev)
        out = torch.empty((B, 1, F, T), dtype=torch.complex64, device=dev)
        parts = (C.c_void_p * len(ins))()
        for j, t_in in enumerate(ins):
            if t_in.dtype != torch.complex64 or t_in.shape != x0.shape:
                raise TypeError("inputs must be complex64 tensors of identical shape")
            parts[j] = L.ptr(torch.view_as_real(t_in))
        tc = None
        if self.cfg.conditional:
            if time_cond is None:
                raise ValueError("time_cond is required for a score network")
            tc = time_cond.to(device=dev, dtype=torch.float32).contiguous()
            if tc.shape != (B,):
                raise ValueError(f"time_cond must have shape [{B}]")
        with self._lock:       # (the ~120 launches of one evaluation go out together: host threads that share this stream's scratch must not interleave)
            L.check(L.lib().storm_ncsnpp_forward(h, parts, len(ins), L.ptr(tc), L.ptr(torch.view_as_real(out)), L.ptr(ws), ws.numel(),
                                                 B, F, T, int(self.negate_output), L.stream()), "storm_ncsnpp_forward")
        return out


    def forward_parts_group(self, ins_list, time_conds=None):
        """P micro-batches of different (B_p, T_p) in ONE C-ABI call (storm_ncsnpp_forward_group): ins_list[p] = the complex channels of
        problem p as contiguous [B_p, F, T_p] tensors, time_conds[p] = float32 [B_p].  Returns the P outputs [B_p, 1, F, T_p].  The op
        sequence is the same for every problem; layers with a grouped kernel run all problems' pixel tiles in one launch."""
        P = len(ins_list)
        if P == 1:
            return [self.forward_parts(ins_list[0], None if time_conds is None else time_conds[0])]
        dev = ins_list[0][0].device
        F = ins_list[0][0].shape[1]
        code = L.dt(self.compute_dtype)
        n_parts = len(ins_list[0])
        Bs, Ts = (C.c_int * P)(), (C.c_int * P)()
        parts, tptr, optr = (C.c_void_p * (P * n_parts))(), (C.c_void_p * P)(), (C.c_void_p * P)()
        outs, keep = [], []
        for p, ins in enumerate(ins_list):
            x0 = ins[0]
            if len(ins) != n_parts or x0.shape[1] != F:
                raise TypeError("every problem needs the same number of complex channels and frequency bins")
            Bs[p], Ts[p] = x0.shape[0], x0.shape[2]
            for j, t_in in enumerate(ins):
                if t_in.dtype != torch.complex64 or t_in.shape != x0.shape or not t_in.is_contiguous():
                    raise TypeError("inputs must be contiguous complex64 tensors of identical shape per problem")
                parts[p * n_parts + j] = L.ptr(torch.view_as_real(t_in))
            if self.cfg.conditional:
                if time_conds is None or time_conds[p] is None:
                    raise ValueError("time_cond is required for a score network")
                tc = time_conds[p].to(device=dev, dtype=torch.float32).contiguous()
                if tc.shape != (x0.shape[0],):
                    raise ValueError(f"time_cond of problem {p} must have shape [{x0.shape[0]}]")
                keep.append(tc)
                tptr[p] = L.ptr(tc)
            out = torch.empty((x0.shape[0], 1, F, x0.shape[2]), dtype=torch.complex64, device=dev)
            outs.append(out)
            optr[p] = L.ptr(torch.view_as_real(out))
        with self._lock:
            h = self._get_handle(code, dev)
            n = L.lib().storm_ncsnpp_group_workspace_bytes(h, P, Bs, Ts, F)
            if n < 0:
                raise L.StormError(f"storm_ncsnpp_group_workspace_bytes: {L.lib().storm_last_error().decode()}")
            key = (str(dev), code, L.stream(), "group")
            ws = self._workspaces.pop(key, None)
            if ws is None or ws.numel() < n:
                ws = None
                ws = torch.empty(n, dtype=torch.uint8, device=dev)
            self._workspaces[key] = ws
            # (launched under the lock: host threads that share this stream's scratch must not interleave the launches of two evaluations)
            L.check(L.lib().storm_ncsnpp_forward_group(h, P, Bs, Ts, F, parts, n_parts, tptr if self.cfg.conditional else None, optr, L.ptr(ws), ws.numel(),
                                                       int(self.negate_output), L.stream()), "storm_ncsnpp_forward_group")
        return outs


@BackboneRegistry.register("ncsnpplarge")
class NCSNppLarge(NCSNpp):
    """~65.6 M parameters (ncsnpp.py:460-470)."""

    def __init__(self, **kwargs):
        for k in ("nf", "ch_mult", "num_res_blocks", "attn_resolutions"):
            kwargs.pop(k, None)
        super().__init__(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), **kwargs)


@BackboneRegistry.register("ncsnpp12M")
class NCSNpp12M(NCSNpp):
    def __init__(self, **kwargs):
        for k in ("nf", "ch_mult", "num_res_blocks", "attn_resolutions"):
            kwargs.pop(k, None)
        super().__init__(nf=96, ch_mult=(1, 2, 2, 1), num_res_blocks=1, attn_resolutions=(0,), **kwargs)


@BackboneRegistry.register("ncsnpp6M")
class NCSNpp6M(NCSNpp):
    def __init__(self, **kwargs):
        for k in ("nf", "ch_mult", "num_res_blocks", "attn_resolutions"):
            kwargs.pop(k, None)
        super().__init__(nf=96, ch_mult=(1, 1, 1, 1), num_res_blocks=1, attn_resolutions=(0,), **kwargs)
