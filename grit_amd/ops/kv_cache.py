"""Key / value cache of step-wise decoding under beam search as one launch per layer and step (grit_kv_append, include/grit_hip.h):
the surviving beams take over their source beam's history and the new token's projected key / value is appended."""
import ctypes

import torch

from grit_amd import lib as _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def append(old_k, old_v, src_beam, new_k, new_v, beam):
    """old_* [B * cur, t, d] (or None at the first step), src_beam [B, beam] int64 or None (identity), new_* [B * beam, 1, d] (last
    dimension dense; may be slices of a wider projection) -> keys, values [B * beam, t + 1, d]."""
    rows, _, d = new_k.shape
    B = rows // beam
    t_old = 0 if old_k is None else old_k.shape[1]
    cur = beam if old_k is None else old_k.shape[0] // B
    if new_k.stride(2) != 1 or new_v.stride(2) != 1 or new_k.stride(0) != new_v.stride(0):
        new_k, new_v = new_k.contiguous(), new_v.contiguous()
    if old_k is not None and not (old_k.is_contiguous() and old_v.is_contiguous()):
        old_k, old_v = old_k.contiguous(), old_v.contiguous()
    if src_beam is not None:
        src_beam = src_beam.reshape(-1).contiguous()
        if src_beam.numel() != rows or src_beam.dtype != torch.int64:
            raise _lib.GritHipError("src_beam must hold one int64 per surviving beam (%d), got %d" % (rows, src_beam.numel()))
    esize = new_k.element_size()
    out_k = torch.empty((rows, t_old + 1, d), dtype=new_k.dtype, device=new_k.device)
    out_v = torch.empty_like(out_k)
    with _lib.device_guard(new_k.device):
        st = _lib.load().grit_kv_append(_ptr(old_k), _ptr(old_v), _ptr(src_beam), B, cur, beam, t_old, d * esize, _ptr(new_k),
                                        _ptr(new_v), new_k.stride(0) * esize, _ptr(out_k), _ptr(out_v), _lib.current_stream_ptr())
    _lib.check(st, "grit_kv_append")
    return out_k, out_v
