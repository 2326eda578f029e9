"""bf16 MFMA GEMM with fused epilogues (grit_gemm_bf16_nt, grit_amd/csrc/gemm.hip) and the Mlp built on it.

`mlp_hidden(x, fc1)` = GELU(fc1(x)) of the Swin Mlp (reference models/common/swin_model.py:31-37) as ONE forward kernel
(GEMM + bias + exact GELU; the pre-activation is kept for the backward) and a backward whose input-gradient GEMM of the
*following* Linear already multiplies by GELU' and sums the bias gradient (`Fc2InputGrad`), so no GELU / GeluBackward /
column-sum kernel touches the [M, 4C] hidden map."""
import ctypes
import os

import torch

from grit_amd import lib as _lib
from grit_amd.ops.profiling import gemm_work, timed

NONE, BIAS, BIAS_GELU, DGELU = 0, 1, 2, 3
# GRIT_GEMM_ROW_SKIP (default 1): the fc2 input gradient skips the tiles of samples that drop path removed from the branch (exact zeros)
ROW_SKIP = os.environ.get("GRIT_GEMM_ROW_SKIP", "1") != "0"
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
    v = VARIANT if variant is None else variant
    work = gemm_work(M, N, K, outputs=2 if (epilogue == BIAS_GELU and aux is not None) else 1,
                     extra_in=1 if epilogue == DGELU else 0)
    with _lib.device_guard(a.device), timed("gemm_own", epilogue=epilogue, kernel="gemm_w4" if v in (7, 9) else ("gemm_short" if 10 <= v <= 13 else "gemm_nt_bf16"), **work):
        st = _lib.load().grit_gemm_bf16_nt(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0), M, N, K,
                                           epilogue, _ptr(bias), _ptr(aux), aux.stride(0) if aux is not None else 0,
                                           _ptr(colsum), VARIANT if variant is None else variant, _lib.current_stream_ptr())
    _lib.check(st, "grit_gemm_bf16_nt")
    return out


BIAS_RELU_DROP, DRELU = 5, 6  # grit_gemm_bf16_nt_relu


def gemm_nt_relu(a, b, epilogue, bias=None, aux=None, p=0.0, seed_dev=None):
    """The decoders' FFN GEMMs on the short-map tiles with ReLU + dropout in the epilogue (grit_gemm_bf16_nt_relu):
    BIAS_RELU_DROP: dropout(relu(a @ b^T + bias), p) -- bit for bit Linear + grit_relu_dropout_fwd;  DRELU: (a @ b^T) with the backward of
    ReLU + dropout applied, aux = the forward's output -- bit for bit the product + grit_relu_dropout_bwd."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    with _lib.device_guard(a.device), timed("gemm_own", epilogue=epilogue, kernel="gemm_short",
                                            **gemm_work(M, N, K, extra_in=1 if epilogue == DRELU else 0)):
        st = _lib.load().grit_gemm_bf16_nt_relu(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0), M, N, K, epilogue,
                                                _ptr(bias), _ptr(aux), aux.stride(0) if aux is not None else 0, float(p),
                                                _ptr(seed_dev), _lib.current_stream_ptr())
    _lib.check(st, "grit_gemm_bf16_nt_relu")
    return out


W4 = 9  # variant of grit_gemm_bf16_nt the long-map policy runs: the persistent four-wave kernel, tile height (256 / 224 rows) by shape


def w4_tile_rows(M, N, device=None):
    """Tile height variant 9 runs for an [M, N] output on the current device (sizes the GELU' column-sum partials: 2 rows per tile row)."""
    return int(_lib.load().grit_gemm_w4_tile_rows(int(M), int(N)))


def gemm_nt_residual(a, b, bias, residual, row_scale=None, rows_per_sample=0, out=None):
    """out[M, N] = residual + row_scale[m // rows_per_sample] * bf16(a @ b^T + bias): the output projection of a Swin branch with its
    residual connection in one launch (grit_gemm_bf16_nt_res); the branch map is never written.  row_scale: [B] float32 or None."""
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    work = gemm_work(M, N, K, extra_in=1)
    with _lib.device_guard(a.device), timed("gemm_own", epilogue=4, kernel="gemm_w4" if (N % 256 == 0 and K % 64 == 0) else "gemm_nt_bf16", **work):
        st = _lib.load().grit_gemm_bf16_nt_res(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0), M, N, K, _ptr(bias),
                                               _ptr(residual), residual.stride(0), _ptr(row_scale) if row_scale is not None else None,
                                               int(rows_per_sample), _lib.current_stream_ptr())
    _lib.check(st, "grit_gemm_bf16_nt_res")
    return out


def _fused_variant(M, N, K):
    """Tile variant of the fused Mlp GEMMs: the four-wave persistent kernel where a tile has >= 16 K steps to amortise its exposed
    GELU epilogue (stage 3, K = 1 024: fc1 + GELU 134 -> 118 us, fc2-dgrad x GELU' 152 -> 134), the eight-wave one elsewhere
    (stage 2: 146 / 186 against 167 / 182; profiles/r04/fused_variants.txt).  GRIT_GEMM_VARIANT overrides."""
    if VARIANT:
        return VARIANT
    return W4 if (OWN and K >= 1024 and K % 64 == 0 and N % 256 == 0 and M >= 256 and M * K * 2 < 2 ** 31 and N * K * 2 < 2 ** 31) else 0


def linear_bias_gelu(x2, weight, bias, row_scale=None, rows_per_sample=0):
    """(pre, act), both [M, N] bf16, one kernel: pre = x2 @ weight^T + bias (what input_grad_dgelu needs), act = gelu(pre).
    With drop-path factors (`row_scale`, one per sample of `rows_per_sample` rows) the tiles of dropped samples are skipped; that path
    always runs the eight-wave kernel, so at K >= 1 024 (stage 3) an active drop path forgoes the four-wave variant's ~12 % there --
    2 of the 22 trainable blocks, against ~11 % of ALL fused-Mlp tiles not computed."""
    M, N = x2.shape[0], weight.shape[0]
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    if (ROW_SKIP and row_scale is not None and rows_per_sample > 0 and row_scale.dtype == torch.float32
            and row_scale.is_cuda):
        # drop path: the tiles of samples whose branch is multiplied by 0 are not computed (zeros in `act` and `pre`)
        act = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
        K = x2.shape[1]
        with _lib.device_guard(x2.device), timed("gemm_own", epilogue=BIAS_GELU, kernel="gemm_nt_bf16", row_scale=row_scale,
                                                  rows_per_sample=int(rows_per_sample), **gemm_work(M, N, K, outputs=2)):
            st = _lib.load().grit_gemm_bf16_nt_rows(_ptr(x2), x2.stride(0), _ptr(weight), weight.stride(0), _ptr(act), act.stride(0), M, N, K,
                                                    BIAS_GELU, _ptr(bias), _ptr(pre), pre.stride(0), None, _ptr(row_scale),
                                                    int(rows_per_sample), VARIANT if VARIANT in (1, 2, 3, 4) else 0, _lib.current_stream_ptr())
        _lib.check(st, "grit_gemm_bf16_nt_rows")
        return pre, act
    act = gemm_nt(x2, weight, BIAS_GELU, bias=bias, aux=pre, variant=_fused_variant(M, N, x2.shape[1]))
    return pre, act


def input_grad_dgelu(dy2, weight_t, pre, row_scale=None, rows_per_sample=0):
    """(d_pre, colsum_partial): d_pre = (dy2 @ weight_t^T) * gelu'(h) with weight_t [N_hidden, K] = the following Linear's
    weight transposed and `pre` = the pre-activation h linear_bias_gelu saved; colsum_partial
    [ceil(M / 128), N_hidden] f32 sums to the bias gradient of the Linear that produced h."""
    M = dy2.shape[0]
    N = weight_t.shape[0]
    if ROW_SKIP and row_scale is not None and rows_per_sample > 0 and row_scale.dtype == torch.float32 and row_scale.is_cuda:
        # drop path: rows of dropped samples are exact zeros in dy2 -- their tiles need no K loop (grit_gemm_bf16_nt_rows, eight-wave kernel)
        partial = torch.empty((-(-M // COLSUM_ROWS), N), dtype=torch.float32, device=dy2.device)
        d_pre = torch.empty((M, N), dtype=torch.bfloat16, device=dy2.device)
        K = dy2.shape[1]
        epi = DGELU
        with _lib.device_guard(dy2.device), timed("gemm_own", epilogue=epi, kernel="gemm_nt_bf16", row_scale=row_scale,
                                                   rows_per_sample=int(rows_per_sample), **gemm_work(M, N, K, extra_in=1)):
            st = _lib.load().grit_gemm_bf16_nt_rows(_ptr(dy2), dy2.stride(0), _ptr(weight_t), weight_t.stride(0), _ptr(d_pre), d_pre.stride(0),
                                                    M, N, K, epi, None, _ptr(pre), pre.stride(0), _ptr(partial), _ptr(row_scale), int(rows_per_sample),
                                                    VARIANT if VARIANT in (1, 2, 3, 4) else 0, _lib.current_stream_ptr())
        _lib.check(st, "grit_gemm_bf16_nt_rows")
        return d_pre, partial
    v = _fused_variant(M, N, dy2.shape[1])
    # (the four-wave kernel writes one row of sums per wave block of its 256- or 224-row tiles: 2 ceil(M / height) rows, all written)
    w4_rows = 256 if v == 7 else (w4_tile_rows(M, N) if v == W4 else 0)
    partial = torch.empty((2 * -(-M // w4_rows) if w4_rows else -(-M // COLSUM_ROWS), N), dtype=torch.float32, device=dy2.device)
    d_pre = gemm_nt(dy2, weight_t, DGELU, aux=pre, colsum=partial, variant=v)
    return d_pre, partial


# ---- the long token maps' plain Linears on the own four-wave kernel (grit_gemm_bf16_nt variant 7, grit_amd/csrc/gemm_w4.hip) ----------
# GRIT_GEMM_OWN (default 1): forward GEMMs y = x W^T + b and input gradients dx = dy W (as NT on W^T, grit_amd/ops/transposed.py) of the
# Swin blocks / value projection go to the persistent 128 x 128-wave-tile kernel where it is at least as fast as the tuned library
# kernel (profiles/r04/w4_vs_lib.txt); the rest -- the K >= 1536 problems with 512 output columns, whose 400 tiles fill 256 CUs 1.56
# times: the library's stream-K kernel has no such quantisation -- stays with the library.
OWN = os.environ.get("GRIT_GEMM_OWN", "1") != "0"
OWN_MIN_ROWS = 8192
_CUS = 256


# GRIT_GEMM_OWN_DEEP (default 1, round 6): also the K >= 1 024 products with <= 512 output columns (fc2 forward, fc1 / qkv input
# gradients of every Swin block) on the own kernel, now that 224-row tiles fill the CUs (profiles/r06/w4_vs_lib_224.txt: 0.91-1.14 x
# the library's stream-K kernel stand-alone, ahead of it in the step: profiles/r06/ab_own_deep.txt).  0: the round-5 policy.
OWN_DEEP = os.environ.get("GRIT_GEMM_OWN_DEEP", "1") != "0"
# GRIT_GEMM_OWN_MAX_TILES (default 16 384): the stacked value projection of the deformable decoder (272 000 x 3 072 x 512: 12 756 tiles) is
# 5 % behind the library's stream-K kernel stand-alone (profiles/r04/w4_vs_lib.txt) and level with it inside the step
# (profiles/r06/ab_own_max_tiles.txt: 46.74 against 46.75 ms) -- level means the own kernel runs it.  8192: the library, as up to round 5.
OWN_MAX_TILES = int(os.environ.get("GRIT_GEMM_OWN_MAX_TILES", "16384"))


def prefers_own(M, N, K):
    """Shape policy, measured on MI355X against the tuned library kernels (profiles/r04/w4_vs_lib.txt, profiles/r06/w4_vs_lib_224.txt):
    the own kernel wins or ties where a tile has few K steps (K <= 512: the library pays a ring fill per tile; e.g. stage-1 proj
    58 -> 42 us, stage-2 qkv 84 -> 77 us), at K = 1 024 with wide outputs and -- with 224-row tiles, round 6 -- on the K >= 1 024
    products with <= 512 output columns (400 / 800 / 200 tiles of 256 rows filled 256 CUs 1.56 / 3.1 / 0.78 times); on the 12 756-tile
    stacked value projection the library is 5 % ahead stand-alone and level inside the step (OWN_MAX_TILES)."""
    if N % 256 or K % 64 or M < OWN_MIN_ROWS or M * K * 2 >= 2 ** 31 or N * K * 2 >= 2 ** 31:
        return False
    tiles = -(-M // 256) * (N // 256)
    return tiles <= OWN_MAX_TILES and (OWN_DEEP or K <= 512 or (K <= 1024 and N >= 1024))


def prefers_own_narrow(M, N, K):
    """The stage-0 maps (819 200 tokens, C = 128): 128 / 384 output columns and K <= 512 are pure HBM streams, where the eight-wave
    kernel of gemm.hip (variant 0: tile chosen by shape) runs at 0.48-0.63 of the 8 TB/s bound against the library's 0.35-0.58
    (profiles/r04/s0_gemms.txt: qkv 298 -> 219 us, proj 115 -> 84, qkv input gradient 225 -> 169, fc1 input gradient 274 -> 210)."""
    return N in (128, 384) and K % 32 == 0 and K <= 512 and M >= 262144 and M * max(N, K) * 2 < 2 ** 31


# GRIT_GEMM_OWN_SHORT (default 1, round 6): the Linears of the two decoders and the grid net (640 .. 4 800 rows) on the 64 x 64 x 64 tiles of
# the per-tile kernel (grit_gemm_bf16_nt variant 12: three workgroups per CU, 600 workgroups for a 4 800 x 512 output where the
# library launches 38-150): 1-2.5 us ahead of the tuned library kernel per call at K <= 512, level at K = 1 024 from 2 048 rows on, behind
# it at K >= 2 048 (profiles/r06/small_gemm.txt).  Forward (+ bias) and, on the transposed copies of grit_amd.ops.transposed, input
# gradients.  0: the library.
OWN_SHORT = os.environ.get("GRIT_GEMM_OWN_SHORT", "1") != "0"
SHORT = 12
# (64: the beam-search steps' 64 .. 320-row Linears too -- config 5's 20 decode steps 11.75 -> 11.08 ms with the same tokens,
# profiles/r06/decode_short_min_rows.txt; 512: the training step's maps only)
SHORT_MIN_ROWS = int(os.environ.get("GRIT_GEMM_SHORT_MIN_ROWS", "64"))


# (A/B: the K = 1 024 gate GEMM and the K = 2 048 FFN product of the beam-search steps on the tile too -- 11.03 against 10.77 ms per decode,
# profiles/r06/decode_short_decode_max_k.txt: they stay with the library)
SHORT_DECODE_MAX_K = int(os.environ.get("GRIT_GEMM_SHORT_DECODE_MAX_K", "512"))


def prefers_own_short(M, N, K):
    """(K <= 1 024 at any row count or K <= 2 048 as well: within 0.03 ms per step of this rule, profiles/r06/ab_short_wide.txt)"""
    return (OWN_SHORT and SHORT_MIN_ROWS <= M < OWN_MIN_ROWS and N % 128 == 0 and K % 64 == 0
            and (K <= 512 or (K <= 1024 and M >= 2048) or (M < 512 and K <= SHORT_DECODE_MAX_K)))


def long_linear(x2, weight, bias):
    """x2 [M, K] @ weight [N, K]^T (+ bias) on the own kernel, or None where the library path is to run."""
    if not (OWN and supported(x2, weight) and (bias is None or (bias.dtype == torch.bfloat16 and bias.data_ptr() % 16 == 0))):
        return None
    M, K = x2.shape
    N = weight.shape[0]
    if prefers_own(M, N, K):
        return gemm_nt(x2, weight, BIAS if bias is not None else NONE, bias=bias, variant=W4)
    if prefers_own_narrow(M, N, K):
        return gemm_nt(x2, weight, BIAS if bias is not None else NONE, bias=bias, variant=0)
    if prefers_own_short(M, N, K):
        return gemm_nt(x2, weight, BIAS if bias is not None else NONE, bias=bias, variant=SHORT)
    return None


def long_linear_residual(x2, weight, bias, residual2, scale, rows_per_sample):
    """residual2 + scale[sample] * (x2 @ weight^T + bias) in ONE launch (gemm_nt_residual) where the own kernel takes the shape, or None.
    scale: [B] float32 drop-path factors or None."""
    if not (OWN and RESIDUAL and bias is not None and supported(x2, weight) and bias.dtype == torch.bfloat16 and bias.data_ptr() % 16 == 0
            and residual2.dtype == torch.bfloat16 and residual2.is_contiguous() and residual2.data_ptr() % 16 == 0
            and residual2.shape == (x2.shape[0], weight.shape[0])
            and (prefers_own(x2.shape[0], weight.shape[0], x2.shape[1])
                 or (RESIDUAL_NARROW and prefers_own_narrow(x2.shape[0], weight.shape[0], x2.shape[1])))):
        return None
    if scale is not None and not (scale.is_cuda and scale.dtype == torch.float32 and scale.is_contiguous() and rows_per_sample >= 256
                                  and scale.numel() * rows_per_sample == x2.shape[0]):
        return None
    return gemm_nt_residual(x2, weight, bias, residual2, scale, rows_per_sample)


# GRIT_GEMM_RESIDUAL (default 1, round 6): proj / fc2 of a Swin block write x = shortcut + factor * branch themselves; the LayerNorm that
# follows reads x only (grit_layernorm_fwd instead of grit_add_layernorm_fwd: one map read instead of two, one written instead of two).
RESIDUAL = os.environ.get("GRIT_GEMM_RESIDUAL", "1") != "0"
# GRIT_GEMM_RESIDUAL_NARROW (default 1): also proj / fc2 of the stage-0 map (128 output columns: the per-tile kernel's 256 x 128 tiles)
RESIDUAL_NARROW = os.environ.get("GRIT_GEMM_RESIDUAL_NARROW", "1") != "0"


def long_input_grad(dy2, weight):
    """dx [M, K_in] = dy2 [M, N_out] @ weight [N_out, K_in] as an NT product on the transposed copy of the weight kept by
    grit_amd.ops.transposed: on the own kernel where the policy prefers it, else the library's NT kernel -- which is 5-25 % faster
    than the NN form torch.mm(dy, W) runs (stage 2: fc1 input gradient 102 -> 89 us, qkv 78 -> 69 us).  None: no copy for the
    weight's current value (the caller runs torch.mm)."""
    if not (OWN and dy2.is_cuda and dy2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16):
        return None
    from grit_amd.ops import transposed
    wt = transposed.lookup(weight)
    if wt is None:
        return None
    if supported(dy2, wt) and prefers_own(dy2.shape[0], wt.shape[0], wt.shape[1]):
        return gemm_nt(dy2, wt, NONE, variant=W4)
    if supported(dy2, wt) and prefers_own_narrow(dy2.shape[0], wt.shape[0], wt.shape[1]):
        return gemm_nt(dy2, wt, NONE, variant=0)
    if supported(dy2, wt) and prefers_own_short(dy2.shape[0], wt.shape[0], wt.shape[1]):
        return gemm_nt(dy2, wt, NONE, variant=SHORT)
    if dy2.shape[0] < OWN_MIN_ROWS:
        return None  # short map outside the policy: the library's NN form on the weight itself (the caller's torch.mm)
    with timed("gemm_lib", **gemm_work(dy2.shape[0], wt.shape[0], wt.shape[1])):
        return torch.nn.functional.linear(dy2, wt)
