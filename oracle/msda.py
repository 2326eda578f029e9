"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  ctypes front-end of oracle/msda_oracle.c (numpy in / out).

Follows models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153 for the host-side contract (outputs
zero-initialised, grads returned as a triple) and ms_deform_im2col_cuda.cuh for the arithmetic.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "msda_oracle.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-std=c99", "-o", _LIB, src, "-lm"])
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lsi, loc, aw):
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    aw = np.ascontiguousarray(aw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and aw.shape == (B, Lq, M, L, P) and shapes.shape == (L, 2)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    return value, shapes, lsi, loc, aw, (B, S, M, D, L, Lq, P), ("f32" if dt == np.float32 else "f64")


def msda_forward(value, shapes, lsi, loc, aw):
    """-> out [B, Lq, M*D] (ms_deform_attn_cuda.cu:54,77)."""
    value, shapes, lsi, loc, aw, dims, suf = _prep(value, shapes, lsi, loc, aw)
    B, S, M, D, L, Lq, P = dims
    out = np.zeros((B, Lq, M * D), dtype=value.dtype)
    getattr(_load(), "msda_oracle_fwd_" + suf)(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw),
                                              *[ctypes.c_int(x) for x in dims], _p(out))
    return out


def msda_backward(value, shapes, lsi, loc, aw, grad_out):
    """-> (grad_value, grad_loc, grad_attn_w) (ms_deform_attn_cuda.cu:121-150)."""
    value, shapes, lsi, loc, aw, dims, suf = _prep(value, shapes, lsi, loc, aw)
    go = np.ascontiguousarray(grad_out, dtype=value.dtype)
    gv, gl, ga = np.zeros_like(value), np.zeros_like(loc), np.zeros_like(aw)
    getattr(_load(), "msda_oracle_bwd_" + suf)(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(go),
                                              *[ctypes.c_int(x) for x in dims], _p(gv), _p(gl), _p(ga))
    return gv, gl, ga
