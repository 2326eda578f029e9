"""Sigmoid-gated merge of the two cross-attentions of a caption decoder layer at inference (reference
models/caption/cap_generator.py:44-56) in three launches -- grit_gate_pack, ONE fc_alpha1 GEMM on the stacked inputs,
grit_gate_fuse -- instead of fourteen, with the composed form's roundings (include/grit_hip.h)."""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from grit_amd import lib as _lib
from grit_amd.ops import backend


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def supported(self_att, enc1, enc2, mask_pad, fc):
    d = self_att.shape[-1]
    return (backend.override() is None and self_att.is_cuda and not torch.is_grad_enabled()
            and self_att.dtype in (torch.bfloat16, torch.float32) and d % 8 == 0
            and enc1.dtype == enc2.dtype == mask_pad.dtype == fc.weight.dtype == self_att.dtype
            and enc1.shape == enc2.shape == self_att.shape and mask_pad.numel() * d == self_att.numel()
            and fc.weight.shape == (d, 2 * d) and fc.bias is not None and not torch.is_autocast_enabled())


def pack(self_att, enc1_raw, enc2_raw, mask_pad):
    """[2R, 2d]: rows (self_att, enc1_raw * mask_pad) then rows (self_att, enc2_raw * mask_pad) -- both inputs of fc_alpha1."""
    d = self_att.shape[-1]
    s2 = self_att.reshape(-1, d).contiguous()
    a = enc1_raw.reshape(-1, d).contiguous()
    b = enc2_raw.reshape(-1, d).contiguous()
    m = mask_pad.reshape(-1).contiguous()
    R = s2.shape[0]
    X = torch.empty((2 * R, 2 * d), dtype=s2.dtype, device=s2.device)
    with _lib.device_guard(s2.device):
        st = _lib.load().grit_gate_pack(_ptr(s2), _ptr(a), _ptr(b), _ptr(m), R, d, int(s2.dtype == torch.bfloat16), _ptr(X),
                                        _lib.current_stream_ptr())
    _lib.check(st, "grit_gate_pack")
    return X


def fuse(enc1_raw, enc2_raw, gates, mask_pad):
    """gates [2R, d] = fc_alpha1(pack(...)) -> ((enc1 * sigmoid(gates[:R]) + enc2 * sigmoid(gates[R:])) / sqrt(2)) * mask_pad
    with enc_i = enc_i_raw * mask_pad, shaped like enc1_raw."""
    d = enc1_raw.shape[-1]
    a = enc1_raw.reshape(-1, d).contiguous()
    b = enc2_raw.reshape(-1, d).contiguous()
    m = mask_pad.reshape(-1).contiguous()
    g = gates.reshape(-1, d).contiguous()
    R = a.shape[0]
    if g.shape[0] != 2 * R:
        raise _lib.GritHipError("gates must hold 2 x %d rows, got %d" % (R, g.shape[0]))
    out = torch.empty_like(a)
    with _lib.device_guard(a.device):
        st = _lib.load().grit_gate_fuse(_ptr(a), _ptr(b), _ptr(g), _ptr(m), R, d, float(np.float32(np.sqrt(2))),
                                        int(a.dtype == torch.bfloat16), _ptr(out), _lib.current_stream_ptr())
    _lib.check(st, "grit_gate_fuse")
    return out.view(enc1_raw.shape)


def gated_merge(self_att, enc1_raw, enc2_raw, mask_pad, fc):
    """((enc1 * sigmoid(fc(cat[self_att, enc1])) + enc2 * sigmoid(fc(cat[self_att, enc2]))) / sqrt(2)) * mask_pad with
    enc_i = enc_i_raw * mask_pad; self_att already masked."""
    from grit_amd.ops.linear import own_or_library_linear
    return fuse(enc1_raw, enc2_raw, own_or_library_linear(pack(self_att, enc1_raw, enc2_raw, mask_pad), fc.weight, fc.bias), mask_pad)
