"""Fused Swin (shifted-)window attention on the token-ordered map (bf16 MFMA, fp32 softmax).

One kernel per direction does what reference models/common/swin_model.py spreads over
SwinTransformerBlock.forward :257-293 (pad to a multiple of the window, roll(-shift), window_partition,
window_reverse, roll(+shift), crop), BasicLayer.forward :424-441 (shift mask, value -100) and
WindowAttention.forward :161-183 (q*scale, q@k^T, + relative-position bias, + mask, softmax, @v):
grit_winattn_fwd_bf16 / grit_winattn_bwd_bf16 (include/grit_hip.h).

Inputs are what the *pointwise* qkv Linear produced on the un-partitioned map, [B, H*W, 3C]; window
padding tokens are synthesised in-kernel from `pad_qkv` (= the Linear's bias, since the reference pads
zeros after norm1).  Gradients: dqkv, d rel_bias (summed over windows, flows on to the bias table through
the gather in WindowAttention.relative_position_bias) and d pad_qkv.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend

WINDOW = 12
HEAD_DIM = 32

# bench.py sets this to a list to collect (kind, start_event, end_event, flops) per launch: HIP events recorded on the
# launch stream right around the kernel (same hook as grit_amd/ops/msda.py)
PROFILE_EVENTS = None


class _Timed(object):

    def __init__(self, kind, flops):
        self.kind, self.flops = kind, flops

    def __enter__(self):
        self.on = PROFILE_EVENTS is not None and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on and PROFILE_EVENTS is not None:
            self.b.record()
            PROFILE_EVENTS.append((self.kind, self.a, self.b, self.flops))
        return False


def hbm_bytes(B, nWh, nWw, num_heads, N, tensors):
    """Compulsory HBM bytes of a launch: `tensors` bf16 [N, head_dim] slices per (window, head) -- forward q, k, v, out (4);
    backward q, k, v, dout, dq, dk, dv (7)."""
    return tensors * N * HEAD_DIM * 2 * B * nWh * nWw * num_heads


def _core_flops(B, nWh, nWw, num_heads, N, products):
    """2*N*N*head_dim flops per product per (window, head): forward QK^T + PV (2), backward S, dP, dV, dK, dQ (5)."""
    return products * 2 * N * N * HEAD_DIM * B * nWh * nWw * num_heads


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


class _WindowAttentionFn(Function):

    @staticmethod
    def forward(ctx, qkv, rel_bias, pad_qkv, mask, H, W, num_heads, window, shift, scale, row_scale=None, pad_owner=None, acc=None):
        B, T, C3 = qkv.shape
        C = C3 // 3
        nWh, nWw = -(-H // window), -(-W // window)
        N = window * window
        out = torch.empty((B, T, C), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B * nWh * nWw, num_heads, N), dtype=torch.float32, device=qkv.device)
        nWm = 0 if mask is None else mask.shape[0]
        fwd = _lib.load().grit_winattn_fwd_f32 if qkv.dtype == torch.float32 else _lib.load().grit_winattn_fwd_bf16
        # (the factors reach the forward kernel only when a backward -- which gets the same factors -- will follow)
        rows = (row_scale is not None and qkv.dtype == torch.bfloat16 and row_scale.is_cuda and row_scale.dtype == torch.float32
                and row_scale.is_contiguous() and row_scale.numel() == B and ctx.needs_input_grad[0])
        with _lib.device_guard(qkv.device), _Timed("fwd" if qkv.dtype == torch.bfloat16 else "fwd_f32",
                                                   _core_flops(B, nWh, nWw, num_heads, N, 2)):
            if rows:
                st = _lib.load().grit_winattn_fwd_bf16_rows(_ptr(qkv), _ptr(rel_bias), _ptr(pad_qkv), _ptr(mask), nWm, B, H, W, C,
                                                            num_heads, window, shift, scale, _ptr(out), _ptr(lse), _ptr(row_scale),
                                                            _lib.current_stream_ptr())
            else:
                st = fwd(_ptr(qkv), _ptr(rel_bias), _ptr(pad_qkv), _ptr(mask), nWm, B, H, W, C,
                                                   num_heads, window, shift, scale, _ptr(out), _ptr(lse),
                                                   _lib.current_stream_ptr())
        _lib.check(st, "grit_winattn_fwd")
        ctx.save_for_backward(qkv, rel_bias, pad_qkv, mask, out, lse)
        ctx.geom = (H, W, num_heads, window, shift, scale)
        # drop path: per-image factors [B] float32 of the attention branch (the caller multiplies the branch by them): the gradient
        # that comes back is zero for images with factor 0 -- the backward kernel does not compute their windows
        ctx.row_scale = row_scale if rows else None
        # pad_owner: the Linear bias parameter whose values pad_qkv holds, when its Linear's backward takes d(pad) into its own bias
        # gradient (grit_amd/ops/linear.py leave_bias_extra); acc: zeroed f32 workspace for d(bias) | d(pad) (one fill per step for all blocks)
        ctx.pad_owner, ctx.acc = pad_owner, acc
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, rel_bias, pad_qkv, mask, out, lse = ctx.saved_tensors
        H, W, num_heads, window, shift, scale = ctx.geom
        B, T, C3 = qkv.shape
        C = C3 // 3
        dout = dout.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        # d(bias) and d(pad) are accumulated across workgroups with float atomics: one zero fill for both
        acc, ctx.acc = ctx.acc, None  # (handed over once: a second backward over the same graph gets a fresh fill)
        if acc is None or acc.numel() != rel_bias.numel() + C3 or acc.device != qkv.device or acc.dtype != torch.float32:
            acc = torch.zeros(rel_bias.numel() + C3, dtype=torch.float32, device=qkv.device)
        dbias, dpad = acc[:rel_bias.numel()].view_as(rel_bias), acc[rel_bias.numel():]
        nWm = 0 if mask is None else mask.shape[0]
        flops = _core_flops(B, -(-H // window), -(-W // window), num_heads, window * window, 5)
        rs = ctx.row_scale
        rows = (rs is not None and qkv.dtype == torch.bfloat16 and rs.is_cuda and rs.dtype == torch.float32 and rs.is_contiguous()
                and rs.numel() == B)
        if rows:
            backend.check_dropped_rows(dout, rs, "window_attention backward")
        bwd = _lib.load().grit_winattn_bwd_f32 if qkv.dtype == torch.float32 else _lib.load().grit_winattn_bwd_bf16
        with _lib.device_guard(qkv.device), _Timed("bwd" if qkv.dtype == torch.bfloat16 else "bwd_f32", flops):
            if rows:
                st = _lib.load().grit_winattn_bwd_bf16_rows(_ptr(qkv), _ptr(rel_bias), _ptr(pad_qkv), _ptr(mask), nWm, _ptr(out),
                                                            _ptr(dout), _ptr(lse), B, H, W, C, num_heads, window, shift, scale,
                                                            _ptr(dqkv), _ptr(dbias), _ptr(dpad), _ptr(rs), _lib.current_stream_ptr())
            else:
                st = bwd(_ptr(qkv), _ptr(rel_bias), _ptr(pad_qkv), _ptr(mask), nWm, _ptr(out),
                                                   _ptr(dout), _ptr(lse), B, H, W, C, num_heads, window, shift, scale,
                                                   _ptr(dqkv), _ptr(dbias), _ptr(dpad), _lib.current_stream_ptr())
        _lib.check(st, "grit_winattn_bwd")
        if ctx.pad_owner is not None and ctx.needs_input_grad[2]:
            from grit_amd.ops.linear import leave_bias_extra
            leave_bias_extra(ctx.pad_owner, dpad)  # joins the qkv Linear's bias-gradient sum: no cast, no add
            return dqkv, dbias, None, None, None, None, None, None, None, None, None, None, None
        return dqkv, dbias, dpad.to(pad_qkv.dtype), None, None, None, None, None, None, None, None, None, None


def window_attention(qkv, rel_bias, pad_qkv, H, W, num_heads, window, shift, scale, mask=None, row_scale=None, pad_owner=None, acc=None):
    """qkv [B, H*W, 3C] (q|k|v, each head-major), rel_bias [nH, N, N] fp32, pad_qkv [3C], optional explicit
    additive mask [nW_mask, N, N] (replaces the analytic shift mask).  Returns [B, H*W, C] in qkv's dtype.
    row_scale: drop-path factors [B] of the branch this attention is part of (see _WindowAttentionFn.forward) or None."""
    ov = backend.override()
    if ov is not None:
        return ov.window_attention(qkv, rel_bias, pad_qkv, H, W, num_heads, window, shift, scale, mask=mask)
    _lib.require_device(qkv, rel_bias, pad_qkv, mask)
    C = qkv.shape[-1] // 3
    if window != WINDOW or C // num_heads != HEAD_DIM:
        raise _lib.GritHipError("fused window attention is built for window 12 / head_dim 32 (GRIT's Swin-B); got "
                                "window %d head_dim %d" % (window, C // num_heads))
    in_dtype = qkv.dtype
    # fp32 tensors keep fp32 storage and arithmetic (the reference's precision: parity path); everything else runs the bf16
    # MFMA kernels
    cdt = torch.float32 if in_dtype in (torch.float32, torch.float64) else torch.bfloat16
    out = _WindowAttentionFn.apply(qkv.to(cdt).contiguous(), rel_bias.float().contiguous(), pad_qkv.to(cdt).contiguous(),
                                   None if mask is None else mask.float().contiguous(), H, W, num_heads, window, shift,
                                   float(scale), row_scale, pad_owner if cdt == torch.bfloat16 else None, acc)
    return out.to(in_dtype)
