"""Fused element-wise glue of the two decoders in the training step (grit_amd/csrc/glue.hip): each function replaces a chain of
4-14 tiny torch launches -- the decoder phase of the step is ~1 100 dependent kernels at the ~6 us launch floor -- by one launch
forward and one backward, with the same fp32 arithmetic in the reference's operation order.  Every function falls back to the
composed torch form when its kernel does not apply (CPU / oracle runs, unusual dtypes); GRIT_FUSED_GLUE=0 forces that (A/B knob).

  sampling_geometry   MSDeformAttn.forward's locations and softmax weights   models/ops/modules/ms_deform_attn.py:97-113
  box_refine          sigmoid(delta + inverse_sigmoid(ref))                  models/detection/det_module.py:40-53
  relu_dropout        dropout(relu(x)) of the position-wise FFNs            det_module.py:302-304, models/common/pos_embed.py:44-48
  gated_merge_train   sigmoid-gated merge of the two cross-attentions       models/caption/cap_generator.py:44-56
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend

ENABLED = os.environ.get("GRIT_FUSED_GLUE", "1") != "0"


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _on_device(*ts):
    return (ENABLED and backend.override() is None and all(t.is_cuda for t in ts) and not torch.is_autocast_enabled()
            and not backend.foreign_capture())


# ----------------------------------------------------------------------------------------------------------------------
class _SamplingGeometryFn(Function):

    @staticmethod
    def forward(ctx, offsets, logits, ref, shapes, M, L, P):
        rows = offsets.shape[0] * offsets.shape[1]
        off2, log2 = offsets.contiguous(), logits.contiguous()
        ref2 = ref.float().contiguous()
        loc = torch.empty((offsets.shape[0], offsets.shape[1], M, L, P, 2), dtype=torch.float32, device=offsets.device)
        aw = torch.empty((offsets.shape[0], offsets.shape[1], M, L, P), dtype=torch.float32, device=offsets.device)
        with _lib.device_guard(offsets.device):
            st = _lib.load().grit_msda_geometry_fwd(_ptr(off2), _ptr(log2), int(off2.dtype == torch.bfloat16), _ptr(ref2), ref2.shape[-1],
                                                    _ptr(shapes), rows, M, L, P, _ptr(loc), _ptr(aw), _lib.current_stream_ptr())
        _lib.check(st, "grit_msda_geometry_fwd")
        ctx.save_for_backward(aw, ref2, shapes)
        ctx.meta = (rows, M, L, P, offsets.dtype, offsets.shape, logits.shape)
        return loc, aw

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loc, g_aw):
        aw, ref2, shapes = ctx.saved_tensors
        rows, M, L, P, dtype, off_shape, log_shape = ctx.meta
        g_loc = torch.zeros_like(aw).unsqueeze(-1).expand(*aw.shape, 2).contiguous() if g_loc is None else g_loc.float().contiguous()
        g_aw = torch.zeros_like(aw) if g_aw is None else g_aw.float().contiguous()
        d_off = torch.empty(off_shape, dtype=dtype, device=aw.device)
        d_log = torch.empty(log_shape, dtype=dtype, device=aw.device)
        with _lib.device_guard(aw.device):
            st = _lib.load().grit_msda_geometry_bwd(_ptr(g_loc), _ptr(g_aw), _ptr(aw), _ptr(ref2), ref2.shape[-1], _ptr(shapes), rows, M,
                                                    L, P, int(dtype == torch.bfloat16), _ptr(d_off), _ptr(d_log),
                                                    _lib.current_stream_ptr())
        _lib.check(st, "grit_msda_geometry_bwd")
        return d_off, d_log, None, None, None, None, None


def sampling_geometry(offsets, logits, reference_points, spatial_shapes, M, L, P):
    """offsets [N, Lq, M*L*P*2], logits [N, Lq, M*L*P] (outputs of sampling_offsets / attention_weights), reference_points
    [N, Lq, L, 2|4] -> (sampling locations [N, Lq, M, L, P, 2], attention weights [N, Lq, M, L, P]), both float32 (float64
    inputs: the composed form in float64).  None when the fused kernel does not apply."""
    if not (_on_device(offsets, logits, reference_points, spatial_shapes) and offsets.dtype == logits.dtype
            and offsets.dtype in (torch.bfloat16, torch.float32) and reference_points.shape[-1] in (2, 4) and L * P <= 64
            and not reference_points.requires_grad and spatial_shapes.dtype == torch.int64 and spatial_shapes.is_contiguous()):
        return None
    return _SamplingGeometryFn.apply(offsets, logits, reference_points, spatial_shapes, M, L, P)


# ----------------------------------------------------------------------------------------------------------------------
def box_refine(delta, reference_points):
    """sigmoid(delta + inverse_sigmoid(reference_points)) (4-d references) / the 2-d form of DetectionModule.bbox_refine, float32,
    no gradient (the caller detaches the result, as the reference does).  None when the fused kernel does not apply."""
    if not (_on_device(delta, reference_points) and delta.dtype in (torch.bfloat16, torch.float32) and delta.shape[-1] == 4
            and reference_points.shape[-1] in (2, 4) and delta.shape[:-1] == reference_points.shape[:-1]):
        return None
    with torch.no_grad():
        d2 = delta.detach().contiguous()
        r2 = reference_points.detach().float().contiguous()
        out = torch.empty(d2.shape, dtype=torch.float32, device=d2.device)
        with _lib.device_guard(d2.device):
            st = _lib.load().grit_box_refine(_ptr(d2), int(d2.dtype == torch.bfloat16), _ptr(r2), r2.shape[-1], d2.numel() // 4, _ptr(out),
                                             _lib.current_stream_ptr())
        _lib.check(st, "grit_box_refine")
    return out


# ----------------------------------------------------------------------------------------------------------------------
class _ReluDropoutFn(Function):

    @staticmethod
    def forward(ctx, x, p, seed_dev):
        x2 = x.contiguous()
        y = torch.empty_like(x2)
        with _lib.device_guard(x.device):
            st = _lib.load().grit_relu_dropout_fwd(_ptr(x2), x2.numel(), float(p), _ptr(seed_dev) if p > 0 else None,
                                                   int(x2.dtype == torch.bfloat16), _ptr(y), _lib.current_stream_ptr())
        _lib.check(st, "grit_relu_dropout_fwd")
        ctx.save_for_backward(x2, seed_dev)
        ctx.p = float(p)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, seed_dev = ctx.saved_tensors
        dy2 = dy.contiguous()
        dx = torch.empty_like(x2)
        with _lib.device_guard(x2.device):
            st = _lib.load().grit_relu_dropout_bwd(_ptr(x2), _ptr(dy2), x2.numel(), ctx.p, _ptr(seed_dev) if ctx.p > 0 else None,
                                                   int(x2.dtype == torch.bfloat16), _ptr(dx), _lib.current_stream_ptr())
        _lib.check(st, "grit_relu_dropout_bwd")
        return dx, None, None


def relu_dropout_backward(y, dy, p, seed_dev):
    """dy with the backward of dropout(relu(.)) applied, from the forward's OUTPUT y (positive exactly where the unit was active and
    kept): the same kernel as _ReluDropoutFn.backward, which only looks at the sign of its first operand and regenerates the keep factor."""
    y2, dy2 = y.contiguous(), dy.contiguous()
    dx = torch.empty_like(dy2)
    with _lib.device_guard(y2.device):
        st = _lib.load().grit_relu_dropout_bwd(_ptr(y2), _ptr(dy2), y2.numel(), float(p), _ptr(seed_dev) if p > 0 else None,
                                               int(y2.dtype == torch.bfloat16), _ptr(dx), _lib.current_stream_ptr())
    _lib.check(st, "grit_relu_dropout_bwd")
    return dx


def relu_dropout(x, p, training):
    """dropout(relu(x), p, training) -- one launch forward, one backward (mask regenerated from a device seed)."""
    p = float(p) if training else 0.0
    if not (_on_device(x) and x.dtype in (torch.bfloat16, torch.float32) and torch.is_grad_enabled() and x.requires_grad
            and 0.0 <= p < 1.0):
        return F.dropout(F.relu(x), p, True) if p > 0 else F.relu(x)
    seed_dev = backend.dropout_seed(x.device) if p > 0 else None
    return _ReluDropoutFn.apply(x, p, seed_dev)


# ----------------------------------------------------------------------------------------------------------------------
class _GatedMergeFn(Function):
    """(self_att, enc1_raw, enc2_raw, mask_pad, W, b) -> ((e1 s(fc[self, e1]) + e2 s(fc[self, e2])) / sqrt 2) * m, e_i = enc_i m."""

    @staticmethod
    def forward(ctx, self_att, enc1, enc2, mask_pad, weight, bias):
        from grit_amd.ops import gate as G
        X = G.pack(self_att, enc1, enc2, mask_pad)
        gates = F.linear(X, weight, bias)
        out = G.fuse(enc1, enc2, gates, mask_pad)
        ctx.save_for_backward(enc1.reshape(-1, enc1.shape[-1]), enc2.reshape(-1, enc2.shape[-1]), mask_pad.reshape(-1), gates, X, weight)
        ctx.shape = self_att.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        from grit_amd.ops.linear import small_weight_bias_grad, column_sum
        enc1, enc2, m, gates, X, weight = ctx.saved_tensors
        d = enc1.shape[-1]
        R = enc1.shape[0]
        enc1, enc2 = enc1.contiguous(), enc2.contiguous()
        d_out2 = d_out.reshape(R, d).contiguous()
        bf = int(d_out2.dtype == torch.bfloat16)
        div = float(np.float32(np.sqrt(2)))
        dG = torch.empty_like(gates)
        lib = _lib.load()
        with _lib.device_guard(d_out2.device):
            st = lib.grit_gate_bwd_a(_ptr(d_out2), _ptr(enc1), _ptr(enc2), _ptr(gates), _ptr(m), R, d, div, bf, _ptr(dG),
                                     _lib.current_stream_ptr())
        _lib.check(st, "grit_gate_bwd_a")
        dX = torch.mm(dG, weight)                      # [2R, 2d]
        dW = torch.mm(dG.t(), X)                       # fc_alpha1 is applied to both branches: ONE gradient for both
        db = column_sum(dG, weight.dtype) if dG.shape[1] % 8 == 0 else dG.sum(0)
        d_self, d_e1, d_e2 = torch.empty_like(enc1), torch.empty_like(enc1), torch.empty_like(enc1)
        with _lib.device_guard(d_out2.device):
            st = lib.grit_gate_bwd_b(_ptr(d_out2), _ptr(gates), _ptr(m), _ptr(dX), R, d, div, bf, _ptr(d_self), _ptr(d_e1), _ptr(d_e2),
                                     _lib.current_stream_ptr())
        _lib.check(st, "grit_gate_bwd_b")
        return d_self.view(ctx.shape), d_e1.view(ctx.shape), d_e2.view(ctx.shape), None, dW, db


def gated_merge_train(self_att, enc1_raw, enc2_raw, mask_pad, fc):
    """Training-time gated merge (same value as the composed form of ParallelAttentionLayer.forward); None when it does not apply."""
    d = self_att.shape[-1]
    if not (_on_device(self_att, enc1_raw, enc2_raw, mask_pad) and torch.is_grad_enabled() and self_att.dtype in (torch.bfloat16, torch.float32)
            and enc1_raw.dtype == enc2_raw.dtype == mask_pad.dtype == fc.weight.dtype == self_att.dtype and d % 8 == 0
            and enc1_raw.shape == enc2_raw.shape == self_att.shape and mask_pad.numel() * d == self_att.numel()
            and fc.weight.shape == (d, 2 * d) and fc.bias is not None):
        return None
    return _GatedMergeFn.apply(self_att, enc1_raw, enc2_raw, mask_pad, fc.weight, fc.bias)
