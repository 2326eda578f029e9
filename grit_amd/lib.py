"""ctypes binding of libgrit_hip.so (C ABI in include/grit_hip.h).

There is no fallback: if the library is missing or a call fails this raises.  Tensors are handed over
as raw device pointers together with torch's *current* HIP stream, so launches are ordered with the
surrounding torch ops (same contract as the reference's at::cuda::getCurrentCUDAStream(),
models/ops/src/cuda/ms_deform_attn_cuda.cu:65).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgrit_hip.so")
ABI_VERSION = 43

_c = ctypes
_ptr, _int, _i64, _f32, _u64 = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_uint64
_ATTN_IN = [_ptr, _i64, _i64] * 3 + [_ptr, _i64, _i64]  # q, k, v (+ row / batch strides), mask (+ strides)

# name -> argument types; mirrors include/grit_hip.h one to one (tests check every symbol resolves)
SIGNATURES = {
    "grit_abi_version": [],
    "grit_status_string": [_int],
    "grit_msda_fwd_f32": [_ptr] * 5 + [_int] * 7 + [_ptr, _ptr],
    "grit_msda_fwd_f64": [_ptr] * 5 + [_int] * 7 + [_ptr, _ptr],
    "grit_msda_bwd_f32": [_ptr] * 6 + [_int] * 7 + [_ptr] * 4,
    "grit_msda_bwd_f64": [_ptr] * 6 + [_int] * 7 + [_ptr] * 4,
    "grit_msda_fwd_bf16": [_ptr] * 5 + [_int] * 7 + [_ptr, _ptr],
    "grit_msda_bwd_bf16": [_ptr] * 6 + [_int] * 7 + [_ptr] * 4,
    "grit_msda_bwd_bf16acc": [_ptr] * 6 + [_int] * 7 + [_ptr] * 4,
    "grit_msda_fwd_bf16_strided": [_ptr, _c.c_long] + [_ptr] * 4 + [_int] * 7 + [_ptr, _ptr],
    "grit_msda_bwd_bf16acc_strided": [_ptr, _c.c_long] + [_ptr] * 5 + [_int] * 7 + [_ptr] * 4,
    "grit_msda_bwd_bf16_staged": [_ptr, _c.c_long] + [_ptr] * 5 + [_int] * 7 + [_ptr] * 6,
    "grit_wgrad_tn_group_ok": [_int] * 3,
    "grit_wgrad_tn_grouped": [_ptr, _int, _ptr],
    "grit_colsum_grouped": [_ptr, _int, _ptr],
    "grit_transpose_bf16_grouped": [_ptr, _int, _ptr],
    "grit_wgrad_tn_splits": [_int] * 3,
    "grit_wgrad_tn": [_ptr, _c.c_long, _ptr, _c.c_long] + [_int] * 4 + [_ptr, _ptr, _ptr],
    "grit_wgrad_tn_rows": [_ptr, _c.c_long, _ptr, _c.c_long] + [_int] * 4 + [_ptr, _ptr, _ptr, _int, _ptr],
    "grit_msda_bwd_sorted_supported": [_int] * 6,
    "grit_msda_bwd_bf16_sorted": [_ptr, _c.c_long] + [_ptr] * 5 + [_int] * 7 + [_ptr] * 4,
    "grit_winattn_fwd_bf16": [_ptr] * 4 + [_int] * 8 + [_f32, _ptr, _ptr, _ptr],
    "grit_winattn_fwd_bf16_rows": [_ptr] * 4 + [_int] * 8 + [_f32, _ptr, _ptr, _ptr, _ptr],
    "grit_winattn_bwd_bf16": [_ptr] * 4 + [_int] + [_ptr] * 3 + [_int] * 7 + [_f32] + [_ptr] * 4,
    "grit_winattn_bwd_bf16_rows": [_ptr] * 4 + [_int] + [_ptr] * 3 + [_int] * 7 + [_f32] + [_ptr] * 5,
    "grit_winattn_fwd_f32": [_ptr] * 4 + [_int] * 8 + [_f32, _ptr, _ptr, _ptr],
    "grit_winattn_bwd_f32": [_ptr] * 4 + [_int] + [_ptr] * 3 + [_int] * 7 + [_f32] + [_ptr] * 4,
    "grit_layernorm_fwd": [_ptr] * 3 + [_int, _int, _f32, _int, _int] + [_ptr] * 4,
    "grit_layernorm_bwd": [_ptr] * 5 + [_int] * 4 + [_ptr] * 4,
    "grit_patch_embed_ln_fwd": [_ptr, _int, _int, _int, _int, _int] + [_ptr] * 4 + [_f32, _ptr, _ptr],
    "grit_merge_layernorm_fwd": [_ptr] + [_int] * 4 + [_ptr] * 2 + [_f32, _int, _int] + [_ptr] * 4,
    "grit_merge_layernorm_bwd": [_ptr] + [_int] * 4 + [_ptr] * 4 + [_int, _int] + [_ptr] * 4,
    "grit_relbias_fwd": [_ptr, _ptr, _int, _int, _int, _int, _ptr, _ptr],
    "grit_relbias_fwd_grouped": [_ptr, _int, _ptr],
    "grit_relbias_bwd_grouped": [_ptr, _int, _ptr],
    "grit_relbias_bwd": [_ptr, _ptr, _ptr, _int, _int, _int, _int, _ptr, _ptr],
    "grit_add_layernorm_fwd": [_ptr] * 3 + [_int, _f32, _ptr] + [_ptr] * 2 + [_int, _int, _f32, _int, _int] + [_ptr] * 5,
    "grit_add_layernorm_bwd": [_ptr] * 7 + [_int, _f32, _ptr] + [_int] * 4 + [_ptr] * 6,
    "grit_groupnorm_tokens_fwd": [_ptr, _c.c_long, _ptr, _ptr, _int, _int, _int, _int, _f32, _int, _int, _ptr, _c.c_long] + [_ptr] * 4,
    "grit_groupnorm_tokens_bwd": [_ptr, _c.c_long, _ptr, _c.c_long, _ptr, _ptr, _ptr, _int, _int, _int, _int, _int, _int] + [_ptr] * 5,
    "grit_adam_flat": [_ptr, _ptr, _int, _ptr, _ptr, _ptr, _c.c_long] + [_f32] * 7 + [_ptr],
    "grit_adam_flat_dev": [_ptr, _ptr, _int, _ptr, _ptr, _ptr, _c.c_long] + [_f32] * 4 + [_ptr, _ptr],
    "grit_resample_taps_bicubic": [_int, _int, _ptr, _ptr, _c.c_long],
    "grit_image_batch_fwd": [_ptr] * 5 + [_int] * 6 + [_ptr] * 3,
    "grit_colsum": [_ptr, _int, _int, _int, _int, _ptr, _ptr],
    "grit_slab_sum": [_ptr, _int, _c.c_long, _int, _c.c_long, _ptr, _int, _ptr],
    "grit_slab_sum_grouped": [_ptr, _int, _ptr],
    "grit_msda_geometry_fwd": [_ptr, _ptr, _int, _ptr, _int, _ptr, _c.c_long, _int, _int, _int, _ptr, _ptr, _ptr],
    "grit_msda_geometry_bwd": [_ptr, _ptr, _ptr, _ptr, _int, _ptr, _c.c_long, _int, _int, _int, _int, _ptr, _ptr, _ptr],
    "grit_box_refine": [_ptr, _int, _ptr, _int, _c.c_long, _ptr, _ptr],
    "grit_relu_dropout_fwd": [_ptr, _c.c_long, _f32, _ptr, _int, _ptr, _ptr],
    "grit_relu_dropout_bwd": [_ptr, _ptr, _c.c_long, _f32, _ptr, _int, _ptr, _ptr],
    "grit_gate_bwd_a": [_ptr, _ptr, _ptr, _ptr, _ptr, _c.c_long, _int, _f32, _int, _ptr, _ptr],
    "grit_gate_bwd_b": [_ptr, _ptr, _ptr, _ptr, _c.c_long, _int, _f32, _int, _ptr, _ptr, _ptr, _ptr],
    "grit_wgrad_small_splits": [_int, _int, _int],
    "grit_wgrad_group_splits": [_int],
    "grit_wgrad_small_grouped": [_ptr, _int, _ptr],
    "grit_wgrad_small": [_ptr, _c.c_long, _ptr, _c.c_long, _int, _int, _int, _int, _ptr, _ptr, _ptr],
    "grit_attn_fwd_f32": _ATTN_IN + [_int] * 5 + [_f32, _f32, _u64, _ptr, _ptr, _ptr, _ptr],
    "grit_attn_fwd_bf16": _ATTN_IN + [_int] * 5 + [_f32, _f32, _u64, _ptr, _ptr, _ptr, _ptr],
    "grit_attn_bwd_f32": _ATTN_IN + [_ptr] * 3 + [_int] * 5 + [_f32, _f32, _u64] + [_ptr] * 5,
    "grit_attn_bwd_bf16": _ATTN_IN + [_ptr] * 3 + [_int] * 5 + [_f32, _f32, _u64] + [_ptr] * 5,
    "grit_topk_rows_f32": [_ptr, _c.c_long, _int, _int, _int, _ptr, _ptr, _ptr],
    "grit_decode_step_inputs": [_ptr, _c.c_int64, _ptr, _int, _ptr, _int, _int, _int, _ptr, _ptr, _int, _int, _ptr, _ptr, _ptr, _ptr],
    "grit_kv_append": [_ptr, _ptr, _ptr, _int, _int, _int, _int, _int, _ptr, _ptr, _c.c_long, _ptr, _ptr, _ptr],
    "grit_gate_pack": [_ptr] * 4 + [_int, _int, _int, _ptr, _ptr],
    "grit_gate_fuse": [_ptr] * 4 + [_int, _int, _f32, _int, _ptr, _ptr],
    "grit_beam_step_workspace": [_int, _int, _int],
    "grit_beam_step_f32": [_ptr, _c.c_long, _ptr, _ptr, _ptr, _int, _int, _int, _int, _int, _int, _ptr, _c.c_long,
                           _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "grit_gemm_bf16_nt": [_ptr, _c.c_long] * 3 + [_int] * 4 + [_ptr, _ptr, _c.c_long, _ptr, _int, _ptr],
    "grit_gemm_w4_tile_rows": [_int, _int],
    "grit_gemm_bf16_nt_res": [_ptr, _c.c_long] * 3 + [_int] * 3 + [_ptr, _ptr, _c.c_long, _ptr, _int, _ptr],
    "grit_gemm_bf16_nt_relu": [_ptr, _c.c_long] * 3 + [_int] * 4 + [_ptr, _ptr, _c.c_long, _c.c_float, _ptr, _ptr],
    "grit_gemm_bf16_nt_rows": [_ptr, _c.c_long] * 3 + [_int] * 4 + [_ptr, _ptr, _c.c_long, _ptr, _ptr, _int, _int, _ptr],
}

SLAB_GROUP_MAX = 48  # GRIT_SLAB_GROUP_MAX
WGRAD_GROUP_MAX = 32  # GRIT_WGRAD_GROUP_MAX


class SlabJob(_c.Structure):
    """grit_slab_job of include/grit_hip.h."""
    _fields_ = [("partial", _c.c_void_p), ("group_stride", _c.c_long), ("groups", _c.c_int), ("slabs", _c.c_int),
                ("n", _c.c_long), ("out", _c.c_void_p), ("out_is_bf16", _c.c_int), ("extra", _c.c_void_p)]


class WgradJob(_c.Structure):
    """grit_wgrad_job of include/grit_hip.h."""
    _fields_ = [("dY", _c.c_void_p), ("ldy", _c.c_long), ("X", _c.c_void_p), ("ldx", _c.c_long), ("M", _c.c_int), ("N", _c.c_int),
                ("K", _c.c_int), ("splits", _c.c_int), ("dW_partial", _c.c_void_p), ("db_partial", _c.c_void_p),
                ("row_scale", _c.c_void_p), ("rows_per_sample", _c.c_int)]


class ColsumJob(_c.Structure):
    """grit_colsum_job of include/grit_hip.h."""
    _fields_ = [("x", _c.c_void_p), ("ld", _c.c_long), ("M", _c.c_int), ("N", _c.c_int), ("slabs", _c.c_int), ("partial", _c.c_void_p)]


class RelbiasJob(_c.Structure):
    """grit_relbias_job of include/grit_hip.h."""
    _fields_ = [("table", _c.c_void_p), ("index", _c.c_void_p), ("bias", _c.c_void_p), ("n_rows", _c.c_int), ("num_heads", _c.c_int),
                ("n_pos", _c.c_int), ("table_is_bf16", _c.c_int)]


class RelbiasBwdJob(_c.Structure):
    """grit_relbias_bwd_job of include/grit_hip.h."""
    _fields_ = [("dbias", _c.c_void_p), ("order", _c.c_void_p), ("offsets", _c.c_void_p), ("dtable", _c.c_void_p), ("n_rows", _c.c_int),
                ("num_heads", _c.c_int), ("n_pos", _c.c_int), ("table_is_bf16", _c.c_int)]


RELBIAS_GROUP_MAX = 32  # GRIT_RELBIAS_GROUP_MAX


class TransposeJob(_c.Structure):
    """grit_transpose_job of include/grit_hip.h."""
    _fields_ = [("src", _c.c_void_p), ("dst", _c.c_void_p), ("rows", _c.c_int), ("cols", _c.c_int)]


COLSUM_GROUP_MAX = 32
TRANSPOSE_GROUP_MAX = 32
_lib = None


class GritHipError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises GritHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GritHipError(
            "libgrit_hip.so not found at %s -- run `python -m grit_amd.build` (there is no CPU or "
            "PyTorch fallback for the GRIT kernels)" % LIB_PATH)
    # The library resolves its HIP runtime to torch's copy by construction (grit_amd/build.py links libamdhip64.so.7 next to it
    # and sets the run path to $ORIGIN), so the import order of torch and this library does not matter.
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = {"grit_status_string": _c.c_char_p, "grit_beam_step_workspace": _c.c_long}.get(name, _int)
    if lib.grit_abi_version() != ABI_VERSION:
        raise GritHipError("libgrit_hip.so ABI %d != binding ABI %d: rebuild" %
                           (lib.grit_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise GritHipError("%s failed: %s" % (what, load().grit_status_string(status).decode()))


class _NoGuard(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """`with device_guard(t.device):` = torch.cuda.device(t.device), but free when that device is already current (the
    one-process-per-GPU case): a training step enters this ~1 000 times and the generic context manager costs ~5 us each."""
    import torch
    index = device.index
    if index is None or index == torch._C._cuda_getDevice():
        return _NO_GUARD
    return torch.cuda.device(device)


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device (raw handle, no Stream object)."""
    import torch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def require_device(*tensors):
    """Reference behaviour: CPU tensors -> error (ms_deform_attn.h:38 'Not implemented on the CPU')."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GritHipError("Not implemented on the CPU: the GRIT kernels need HIP device tensors")
