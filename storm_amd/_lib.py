"""ctypes binding to libstorm_hip.so (the C ABI declared in include/storm_hip.h).

The product path has no CPU fallback: importing an op without the built library, or
handing it a tensor that is not on a HIP device, raises.  (CPU tests load the lane-accurate
host simulation of the same kernel sources through ``_load_for_tests`` — test infrastructure,
see tests/sim/.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# STORM_LIB: tools/ point this at the profiling build (libstorm_hip_prof.so, python -m storm_amd.build --profiling)
LIB_PATH = os.environ.get("STORM_LIB") or os.path.join(_HERE, "csrc", "libstorm_hip.so")

F32, BF16, F16 = 0, 1, 2
_TORCH2DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_DT2TORCH = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}

OP_NPTR, OP_NINT, OP_NFLT = 13, 24, 4
ABI_VERSION = 2                  # include/storm_hip.h: STORM_ABI_VERSION


class StormError(RuntimeError):
    pass


class ConvSeg(C.Structure):
    _fields_ = [("src_a", C.c_void_p), ("src_b", C.c_void_p), ("Ca", C.c_int), ("Cb", C.c_int),
                ("bstride_a", C.c_longlong), ("bstride_b", C.c_longlong), ("w", C.c_void_p),
                ("CinP", C.c_int), ("w_rows", C.c_int), ("ntaps", C.c_int),
                ("w_bstride", C.c_longlong), ("w_tapstride", C.c_longlong), ("gn_ss", C.c_void_p),
                ("gn_silu", C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [("seg", ConvSeg * 2), ("nseg", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("out", C.c_void_p), ("outC", C.c_int), ("Cout", C.c_int), ("out_bstride", C.c_longlong),
                ("bias", C.c_void_p), ("tbias", C.c_void_p), ("tbias_stride", C.c_int),
                ("skip", C.c_void_p), ("skip_bstride", C.c_longlong), ("scale", C.c_float),
                ("out_f32", C.c_int), ("dtype", C.c_int), ("gn_part", C.c_void_p),
                ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_longlong)]


class NcsnppConfig(C.Structure):
    _fields_ = [("nf", C.c_int), ("n_levels", C.c_int), ("ch_mult", C.c_int * 8), ("num_res_blocks", C.c_int), ("n_attn", C.c_int),
                ("attn_resolutions", C.c_int * 4), ("image_size", C.c_int), ("input_channels", C.c_int), ("discriminative", C.c_int)]


class Ouve(C.Structure):
    _fields_ = [("theta", C.c_float), ("sigma_min", C.c_float), ("sigma_max", C.c_float), ("N", C.c_int)]


class Ref(C.Structure):
    _fields_ = [("buf", C.c_int32), ("pad_", C.c_int32), ("off", C.c_int64)]


class Op(C.Structure):
    _fields_ = [("code", C.c_int32), ("pad_", C.c_int32), ("p", Ref * OP_NPTR),
                ("i", C.c_int64 * OP_NINT), ("f", C.c_float * OP_NFLT)]


_vp, _i, _ll, _f, _u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint64
_SIGNATURES = {
    "storm_abi_version": ([], C.c_int),
    "storm_abi_struct_bytes": ([_i], C.c_longlong),
    "storm_set_switch": ([C.c_char_p, _ll], C.c_int),
    "storm_get_switch": ([C.c_char_p], C.c_longlong),
    "storm_device_info": ([C.c_char_p, _i, C.POINTER(C.c_int), C.POINTER(C.c_size_t)], C.c_int),
    "storm_pack_conv_weight": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_pack_matrix": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_conv": ([C.POINTER(ConvArgs), _vp], C.c_int),
    "storm_conv_group_blob_bytes": ([C.POINTER(ConvArgs), _i], C.c_longlong),
    "storm_conv_group": ([C.POINTER(ConvArgs), _i, _vp, _ll, _i, _vp], C.c_int),
    "storm_conv_tiles": ([C.POINTER(ConvArgs)], C.c_int),
    "storm_conv_splitk_bytes": ([C.POINTER(ConvArgs)], C.c_longlong),
    "storm_conv_kernel_name": ([C.POINTER(ConvArgs)], C.c_char_p),
    "storm_gn_apply_kernel_name": ([_i, _i, _i, _i, _i, _i, _i], C.c_char_p),
    "storm_gn_finalize": ([_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp], C.c_int),
    "storm_gn_finalize_ss": ([_vp, _i, _i, _vp, _i, _i, _i, _i, _ll, _vp, _vp, _f, _vp, _vp, _vp], C.c_int),
    "storm_gn_stats": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp], C.c_int),
    "storm_gn_apply": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _vp], C.c_int),
    "storm_fir_up2": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_fir_down2": ([_vp, _vp, _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_upfirdn2d": ([_vp, _vp, _vp] + [_i] * 14 + [_vp], C.c_int),
    "storm_upfirdn2d_out_size": ([_i] * 6, C.c_longlong),
    "storm_attention_supported": ([_i, _i], C.c_int),
    "storm_attention": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _f, _i, _vp], C.c_int),
    "storm_attention_scratch_bytes": ([_i, _i, _i, _i], C.c_longlong),
    "storm_attention_ws": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _f, _i, _vp, _ll, _vp], C.c_int),
    "storm_attention_group_blob_bytes": ([C.POINTER(C.c_int), C.POINTER(C.c_int), _i], C.c_longlong),
    "storm_attention_group": ([C.POINTER(_vp)] * 4 + [C.POINTER(C.c_int)] * 3 + [_i, _vp, _i, _f, _i, _vp, _ll, _vp], C.c_int),
    "storm_softmax_rows": ([_vp, _vp, _ll, _i, _i, _i, _vp], C.c_int),
    "storm_pack_input": ([C.POINTER(_vp), _i, _vp, _i, _i, _i, _i, _vp], C.c_int),
    "storm_time_embedding": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp], C.c_int),
    "storm_dense": ([_vp, _vp, _vp, _vp, _i, _i, _i, _vp], C.c_int),
    "storm_output_head": ([_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_input_pyramid": ([C.POINTER(_vp), _i, C.POINTER(_vp), _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_output_pyramid": ([C.POINTER(_vp), _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp], C.c_int),
    "storm_ouve_prior": ([_vp, _vp, _vp, _i, _ll, Ouve, _u64, _u64, _vp], C.c_int),
    "storm_ouve_ald_step": ([_vp, _vp, _vp, _vp, _vp, _i, _ll, Ouve, _f, _u64, _u64, _vp], C.c_int),
    "storm_ouve_predictor_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, Ouve, _i, _i, _u64, _u64, _vp], C.c_int),
    "storm_batch_l2norm": ([_vp, _vp, _i, _ll, _vp], C.c_int),
    "storm_langevin_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _f, _i, _vp], C.c_int),
    "storm_si_sdr": ([_vp, _vp, _vp, _i, _ll, _ll, _ll, _f, _vp], C.c_int),
    "storm_ouve_pf_drift": ([_vp, _vp, _vp, _vp, _vp, _i, _ll, Ouve, _vp], C.c_int),
    "storm_ouve_pf_drift_g": ([_vp, _vp, _vp, _vp, _vp, _i, _ll, _f, _vp], C.c_int),
    "storm_sde_prior_rows": ([_vp, _vp, _vp, _vp, _i, _ll, _u64, _u64, _vp], C.c_int),
    "storm_sde_predictor_step_rows": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, _i, _u64, _u64, _vp], C.c_int),
    "storm_sde_pf_drift_rows": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp], C.c_int),
    "storm_rk_combine": ([_vp, _vp, C.POINTER(_vp), C.POINTER(C.c_float), _i, _f, _ll, _vp], C.c_int),
    "storm_rk_scaled_sumsq": ([_vp, _vp, _i, _vp, _vp, C.POINTER(_vp), C.POINTER(C.c_float), _i, _f, _f, _f, _ll, _vp], C.c_int),
    "storm_rk_combine_rows": ([_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(C.c_double), _i, C.POINTER(C.c_double), _i, _ll, _vp], C.c_int),
    "storm_rk_scaled_sumsq_rows": ([_vp, _vp, _ll, _vp, _vp, C.POINTER(_vp), C.POINTER(C.c_double), _i, C.POINTER(C.c_double), C.c_double, C.c_double, _i, _ll, _vp], C.c_int),
    "storm_copy_rows": ([_vp, _vp, C.POINTER(C.c_int), _i, _ll, _vp], C.c_int),
    "storm_complex_randn": ([_vp, _ll, _u64, _u64, _vp], C.c_int),
    "storm_spec_transform": ([_vp, _vp, _ll, _f, _f, _i, _vp], C.c_int),
    "storm_peak_abs": ([_vp, _vp, _i, _ll, _ll, _vp, _vp], C.c_int),
    "storm_stft": ([_vp, _vp, _vp, _vp, _vp, _i, _ll, _ll, _i, _i, _i, _i, _f, _f, _vp, _vp], C.c_int),
    "storm_istft": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _ll, _i, _i, _f, _f, _vp, _vp], C.c_int),
    "storm_program_run": ([C.POINTER(Op), _i, C.POINTER(_vp), _i, _i, _vp], C.c_int),
    "storm_ncsnpp_num_tensors": ([C.POINTER(NcsnppConfig)], C.c_int),
    "storm_ncsnpp_tensor_info": ([C.POINTER(NcsnppConfig), _i, C.c_char_p, _i, C.POINTER(C.c_int), C.POINTER(C.c_longlong)], C.c_int),
    "storm_ncsnpp_arena_bytes": ([C.POINTER(NcsnppConfig), _i], C.c_longlong),
    "storm_ncsnpp_create": ([C.POINTER(NcsnppConfig), C.POINTER(_vp), _i, _i, _vp, _vp, C.POINTER(_vp)], C.c_int),
    "storm_ncsnpp_destroy": ([_vp], None),
    "storm_ncsnpp_set_fusion": ([_vp, _i, _i, _i], C.c_int),
    "storm_ncsnpp_set_graph": ([_vp, _i], C.c_int),
    "storm_ncsnpp_graph_launches": ([_vp], C.c_longlong),
    "storm_ncsnpp_workspace_bytes": ([_vp, _i, _i, _i], C.c_longlong),
    "storm_ncsnpp_forward": ([_vp, C.POINTER(_vp), _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp], C.c_int),
    "storm_ncsnpp_group_workspace_bytes": ([_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _i], C.c_longlong),
    "storm_ncsnpp_forward_group": ([_vp, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _i, C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_vp), _vp, _ll, _i, _vp], C.c_int),
    "storm_ncsnpp_group_launches": ([_vp], C.c_longlong),
    "storm_ncsnpp_program": ([_vp, _i, _i, _i, C.POINTER(C.POINTER(Op)), C.POINTER(C.c_int), C.POINTER(C.c_longlong)], C.c_int),
    "storm_ncsnpp_release_program": ([_vp, C.POINTER(Op)], C.c_int),
    "storm_ncsnpp_arena": ([_vp], _vp),
    "storm_program_kernel_name": ([C.POINTER(Op), _i, _i], C.c_char_p),
    "storm_program_run_timed": ([C.POINTER(Op), _i, C.POINTER(_vp), _i, _i, _vp, C.POINTER(C.c_float)], C.c_int),
}
EXPORTS = ["storm_last_error"] + list(_SIGNATURES)

_lib = None
_sim = False


def _bind(path, hold_gil=False):
    # hold_gil (the test simulator only): its fiber runtime is one global machine - host threads (grouped micro-batches) must enter it one at a time
    lib = C.PyDLL(path) if hold_gil else C.CDLL(path)
    lib.storm_last_error.restype = C.c_char_p
    lib.storm_last_error.argtypes = []
    for name, (args, res) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    # the structures above mirror include/storm_hip.h by hand: refuse a library built from another header
    if lib.storm_abi_version() != ABI_VERSION:
        raise StormError(f"{path}: ABI version {lib.storm_abi_version()}, this binding is written for {ABI_VERSION}")
    for which, cls in ((0, ConvArgs), (1, Op), (2, ConvSeg), (3, NcsnppConfig)):
        if lib.storm_abi_struct_bytes(which) != C.sizeof(cls):
            raise StormError(f"{path}: sizeof({cls.__name__}) = {C.sizeof(cls)} here, {lib.storm_abi_struct_bytes(which)} in the library")
    return lib


def lib():
    """The loaded library; raises if libstorm_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StormError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or python -m storm_amd.build). "
                "storm_amd has no CPU fallback.")
        _lib = _bind(LIB_PATH)
    return _lib


def _load_for_tests(path, sim):
    """TEST HOOK: bind another build of the same C ABI (the host simulation)."""
    global _lib, _sim
    _lib = _bind(path, hold_gil=bool(sim))
    _sim = bool(sim)
    return _lib


def is_sim():
    return _sim


def check(rc, what=""):
    if rc != 0:
        raise StormError(f"{what} failed ({rc}): {lib().storm_last_error().decode()}")


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _TORCH2DT[d]
    except KeyError:
        raise StormError(f"unsupported activation dtype {d}")


def torch_dtype(code):
    return _DT2TORCH[code]


def ptr(t, allow_none=True):
    """Raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        if allow_none:
            return None
        raise StormError("null tensor")
    if not t.is_contiguous():
        raise StormError("tensor must be contiguous")
    if not _sim and not t.is_cuda:
        raise StormError("storm_amd ops run on the GPU only: got a CPU tensor (there is no CPU fallback)")
    return t.data_ptr()


def ptr_rows(t):
    """Pointer of a 2-D tensor whose rows are contiguous (row stride may exceed the row length)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise StormError("expected a row-contiguous 2-D tensor")
    if not _sim and not t.is_cuda:
        raise StormError("storm_amd ops run on the GPU only: got a CPU tensor (there is no CPU fallback)")
    return t.data_ptr()


def stream():
    if _sim:
        return None
    return torch.cuda.current_stream().cuda_stream
