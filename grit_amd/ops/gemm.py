"""bf16 MFMA GEMM with fused epilogues (grit_gemm_bf16_nt, grit_amd/csrc/gemm.hip) and the Mlp built on it.

`mlp_hidden(x, fc1)` = GELU(fc1(x)) of the Swin Mlp (reference models/common/swin_model.py:31-37) as ONE forward kernel
(GEMM + bias + exact GELU; the pre-activation is kept for the backward) and a backward whose input-gradient GEMM of the
*following* Linear already multiplies by GELU' and sums the bias gradient (`Fc2InputGrad`), so no GELU / GeluBackward /
column-sum kernel touches the [M, 4C] hidden map."""
import ctypes
import os

import torch

from grit_amd import lib as _lib
from grit_amd.ops.profiling import timed

NONE, BIAS, BIAS_GELU, DGELU = 0, 1, 2, 3
COLSUM_ROWS = 128
VARIANT = int(os.environ.get("GRIT_GEMM_VARIANT", "0"))  # tuning alternatives of the same kernel (A/B runs)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def supported(x2, weight):
    """[M, K] bf16 x [N, K] bf16, both row-major with 16-byte aligned rows, N % 128 == 0, K % 32 == 0."""
    return (x2.is_cuda and x2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x2.dim() == 2
            and weight.dim() == 2 and x2.shape[1] == weight.shape[1] and weight.shape[0] % 128 == 0
            and weight.shape[1] % 32 == 0 and x2.stride(1) == 1 and weight.stride(1) == 1
            and x2.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)


def gemm_nt(a, b, epilogue=NONE, bias=None, aux=None, colsum=None, out=None, variant=None):
    """out[M, N] = epilogue(a[M, K] @ b[N, K]^T); see include/grit_hip.h for the epilogues."""
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    with _lib.device_guard(a.device), timed("gemm_own", flops=2.0 * M * N * K, epilogue=epilogue):
        st = _lib.load().grit_gemm_bf16_nt(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0), M, N, K,
                                           epilogue, _ptr(bias), _ptr(aux), aux.stride(0) if aux is not None else 0,
                                           _ptr(colsum), VARIANT if variant is None else variant, _lib.current_stream_ptr())
    _lib.check(st, "grit_gemm_bf16_nt")
    return out


def linear_bias_gelu(x2, weight, bias):
    """(pre, act) = (x2 @ weight^T + bias, gelu(x2 @ weight^T + bias)), both [M, N] bf16, one kernel."""
    M, N = x2.shape[0], weight.shape[0]
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    act = gemm_nt(x2, weight, BIAS_GELU, bias=bias, aux=pre)
    return pre, act


def input_grad_dgelu(dy2, weight_t, pre):
    """(d_pre, colsum_partial): d_pre = (dy2 @ weight_t^T) * gelu'(pre) with weight_t [N_hidden, K] = the following Linear's
    weight transposed; colsum_partial [ceil(M / 128), N_hidden] f32 sums to the bias gradient of the Linear that produced pre."""
    M = dy2.shape[0]
    N = weight_t.shape[0]
    partial = torch.empty((-(-M // COLSUM_ROWS), N), dtype=torch.float32, device=dy2.device)
    d_pre = gemm_nt(dy2, weight_t, DGELU, aux=pre, colsum=partial)
    return d_pre, partial
