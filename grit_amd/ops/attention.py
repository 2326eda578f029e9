"""Fused scaled-dot attention core (fp32 arithmetic) for the caption decoder, the grid network and the
150-query self-attention of the deformable decoder layers.

Replaces the chain of reference models/common/attention.py:71-84
    scores = q @ k^T / sqrt(d_k); scores.masked_fill(mask, -inf); softmax; dropout; @ v
(and the same chain inside nn.MultiheadAttention, models/detection/det_module.py:330-333) with one kernel
launch per direction: grit_attn_fwd_* / grit_attn_bwd_* (include/grit_hip.h).  Softmax statistics are
kept per row (log-sum-exp) so the backward recomputes P instead of storing [B,H,Tq,Nk].
Dropout on P uses a counter-based hash RNG keyed by (seed, b, h, q, k): the backward regenerates the same
keep-mask from the seed saved in the autograd context.
"""
import ctypes
import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend

HEAD_DIM = 64  # d_k of every attention on GRIT's path (d_model 512 / 8 heads)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _rows(t):
    """[B, T, H, D] view whose (H, D) block is dense -> (tensor, row stride in elements, batch stride)."""
    if t.stride(3) != 1 or t.stride(2) != t.shape[3]:
        t = t.contiguous()
    return t, t.stride(1), t.stride(0)


def _mask_args(mask, B, Tq, Nk):
    """bool mask broadcastable to [B,1,Tq,Nk] -> (uint8 tensor, batch stride, query stride)."""
    if mask is None:
        return None, 0, 0
    m = mask
    if m.dim() != 4 or m.shape[1] != 1 or m.shape[3] != Nk or m.shape[0] not in (1, B) or m.shape[2] not in (1, Tq):
        raise RuntimeError("attention mask must be [B|1, 1, Tq|1, Nk], got %s" % (tuple(m.shape),))
    # a bool tensor is one byte per element holding 0 / 1: reinterpreted in place (no conversion launch per attention call)
    m = m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8).contiguous()
    sb = 0 if m.shape[0] == 1 else m.shape[2] * Nk
    sq = 0 if m.shape[2] == 1 else Nk
    return m, sb, sq


class _AttentionFn(Function):

    @staticmethod
    def forward(ctx, q, k, v, mask, scale, dropout_p, seed, seed_dev=None):
        B, Tq, H, D = q.shape
        Nk = k.shape[1]
        q, ldq, bsq = _rows(q)
        k, ldk, bsk = _rows(k)
        v, ldv, bsv = _rows(v)
        m, msb, msq = _mask_args(mask, B, Tq, Nk)
        out = torch.empty((B, Tq, H * D), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
        L = _lib.load()
        fn = L.grit_attn_fwd_bf16 if q.dtype == torch.bfloat16 else L.grit_attn_fwd_f32
        with _lib.device_guard(q.device):
            st = fn(_ptr(q), ldq, bsq, _ptr(k), ldk, bsk, _ptr(v), ldv, bsv, _ptr(m), msb, msq, B, H, Tq, Nk, D,
                    scale, dropout_p, seed, _ptr(seed_dev), _ptr(out), _ptr(lse), _lib.current_stream_ptr())
        _lib.check(st, "grit_attn_fwd")
        ctx.save_for_backward(q, k, v, m, out, lse, seed_dev)
        ctx.args = (scale, dropout_p, seed, msb, msq)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, m, out, lse, seed_dev = ctx.saved_tensors
        scale, dropout_p, seed, msb, msq = ctx.args
        B, Tq, H, D = q.shape
        Nk = k.shape[1]
        dout = dout.contiguous()
        dq = torch.empty((B, Tq, H, D), dtype=q.dtype, device=q.device)
        dk = torch.empty((B, Nk, H, D), dtype=q.dtype, device=q.device)
        dv = torch.empty((B, Nk, H, D), dtype=q.dtype, device=q.device)
        L = _lib.load()
        fn = L.grit_attn_bwd_bf16 if q.dtype == torch.bfloat16 else L.grit_attn_bwd_f32
        with _lib.device_guard(q.device):
            st = fn(_ptr(q), q.stride(1), q.stride(0), _ptr(k), k.stride(1), k.stride(0), _ptr(v), v.stride(1),
                    v.stride(0), _ptr(m), msb, msq, _ptr(out), _ptr(dout), _ptr(lse), B, H, Tq, Nk, D, scale,
                    dropout_p, seed, _ptr(seed_dev), _ptr(dq), _ptr(dk), _ptr(dv), _lib.current_stream_ptr())
        _lib.check(st, "grit_attn_bwd")
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, mask=None, scale=None, dropout_p=0.0, training=False):
    """q [B,Tq,H,64], k/v [B,Nk,H,64] (float32 or bfloat16), mask bool [B|1,1,Tq|1,Nk] (True = masked) -> [B,Tq,H*64]."""
    ov = backend.override()
    if ov is not None:
        return ov.attention(q, k, v, mask, scale=scale, dropout_p=dropout_p, training=training)
    _lib.require_device(q, k, v, mask)
    if q.shape[-1] != HEAD_DIM:
        raise _lib.GritHipError("fused attention supports head_dim 64 only (got %d)" % q.shape[-1])
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if q.dtype not in (torch.float32, torch.bfloat16):
        q = q.float()
    k, v = k.to(q.dtype), v.to(q.dtype)
    p = float(dropout_p) if training else 0.0
    # the dropout seed lives in device memory and is drawn by torch's device generator: no host sync
    seed_dev = backend.dropout_seed(q.device) if p > 0 else None
    return _AttentionFn.apply(q, k, v, mask, float(scale), p, 0, seed_dev)
