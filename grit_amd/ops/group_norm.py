"""GroupNorm of the detector's projected feature levels, token-major, written straight into the flat multi-level map.

    flat[:, start_l : start_l + T_l, :] = GroupNorm(G)(x_l)      x_l [B, T_l, C] = 1x1 conv (a GEMM) of feature level l

One autograd node for all levels: forward = 2 launches per level (grit_groupnorm_tokens_fwd), no NCHW permute, no
torch.cat; backward reads the gradient of the flat map slice by slice.  Reference: models/caption/detector.py:28-33,58
(Conv2d 1x1 + GroupNorm(32, hidden_dim) per level) and models/detection/det_module.py:172-175 (flatten + cat)."""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend

GN_CHUNKS = 16  # GRIT_GN_CHUNKS in include/grit_hip.h


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class _GroupNormLevelsFn(Function):

    @staticmethod
    def forward(ctx, G, eps, n_levels, *args):
        xs, ws, bs = args[:n_levels], args[n_levels:2 * n_levels], args[2 * n_levels:]
        B, C = xs[0].shape[0], xs[0].shape[2]
        Ts = [x.shape[1] for x in xs]
        S = sum(Ts)
        flat = torch.empty(B, S, C, dtype=xs[0].dtype, device=xs[0].device)
        stats = torch.empty(n_levels, 2, B, G, dtype=torch.float32, device=flat.device)
        work = torch.empty(B * GN_CHUNKS * 2 * G, dtype=torch.float32, device=flat.device)
        xb, wb = int(flat.dtype == torch.bfloat16), int(ws[0].dtype == torch.bfloat16)
        lib = _lib.load()
        start = 0
        with _lib.device_guard(flat.device):
            stream = _lib.current_stream_ptr()
            for l in range(n_levels):
                out = flat[:, start:start + Ts[l]]
                st = lib.grit_groupnorm_tokens_fwd(_ptr(xs[l]), xs[l].stride(0), _ptr(ws[l]), _ptr(bs[l]), B, Ts[l], C, G, eps,
                                                   xb, wb, _ptr(out), flat.stride(0), _ptr(stats[l, 0]), _ptr(stats[l, 1]),
                                                   _ptr(work), stream)
                _lib.check(st, "grit_groupnorm_tokens_fwd")
                start += Ts[l]
        ctx.save_for_backward(stats, *xs, *ws)
        ctx.meta = (G, n_levels, Ts, B, C)
        return flat

    @staticmethod
    @once_differentiable
    def backward(ctx, dflat):
        G, n_levels, Ts, B, C = ctx.meta
        stats = ctx.saved_tensors[0]
        xs, ws = ctx.saved_tensors[1:1 + n_levels], ctx.saved_tensors[1 + n_levels:]
        if dflat.stride(2) != 1 or dflat.stride(1) != C or dflat.dtype != xs[0].dtype:
            dflat = dflat.to(xs[0].dtype).contiguous()
        work = torch.empty(B * GN_CHUNKS * 2 * C, dtype=torch.float32, device=dflat.device)
        xb, wb = int(dflat.dtype == torch.bfloat16), int(ws[0].dtype == torch.bfloat16)
        lib = _lib.load()
        dxs, dws, dbs = [], [], []
        start = 0
        with _lib.device_guard(dflat.device):
            stream = _lib.current_stream_ptr()
            for l in range(n_levels):
                dy = dflat[:, start:start + Ts[l]]
                dx = torch.empty(B, Ts[l], C, dtype=dflat.dtype, device=dflat.device)
                dwb = torch.empty(2, C, dtype=ws[l].dtype, device=dflat.device)
                st = lib.grit_groupnorm_tokens_bwd(_ptr(xs[l]), xs[l].stride(0), _ptr(dy), dflat.stride(0), _ptr(ws[l]),
                                                   _ptr(stats[l, 0]), _ptr(stats[l, 1]), B, Ts[l], C, G, xb, wb, _ptr(dx),
                                                   _ptr(dwb[0]), _ptr(dwb[1]), _ptr(work), stream)
                _lib.check(st, "grit_groupnorm_tokens_bwd")
                dxs.append(dx); dws.append(dwb[0]); dbs.append(dwb[1])
                start += Ts[l]
        return (None, None, None, *dxs, *dws, *dbs)


def _fits(xs, ws, bs, G):
    x0 = xs[0]
    if backend.override() is not None or not x0.is_cuda or torch.is_autocast_enabled():
        return False
    C = x0.shape[-1]
    if C not in (256, 512) or C % G or (C // G) % 8 or ((C // G) // 8) & ((C // G) // 8 - 1) or G > 64:
        return False
    for x, w, b in zip(xs, ws, bs):
        if (x.dim() != 3 or x.shape[0] != x0.shape[0] or x.shape[2] != C or x.dtype != x0.dtype or x.stride(2) != 1
                or x.stride(1) != C or x.stride(0) % 8 or w.dtype != ws[0].dtype or b.dtype != w.dtype or not w.is_contiguous()
                or not b.is_contiguous()):
            return False
    if x0.dtype not in (torch.float32, torch.bfloat16):
        return False
    return ws[0].dtype == x0.dtype or (x0.dtype == torch.bfloat16 and ws[0].dtype == torch.float32)


def group_norm_levels(xs, weights, biases, num_groups, eps=1e-5):
    """xs: list of [B, T_l, C] token maps -> [B, sum T_l, C], level l normalised with (weights[l], biases[l])."""
    xs, weights, biases = list(xs), list(weights), list(biases)
    if _fits(xs, weights, biases, num_groups):
        return _GroupNormLevelsFn.apply(num_groups, float(eps), len(xs), *xs, *weights, *biases)
    # composition with identical semantics: GroupNorm on the channels-first view of each level, then concatenation
    outs = [F.group_norm(x.transpose(1, 2), num_groups, w, b, eps).transpose(1, 2) for x, w, b in zip(xs, weights, biases)]
    return torch.cat(outs, 1)
