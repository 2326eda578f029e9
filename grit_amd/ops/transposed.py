"""Transposed copies of weights, made for many layers in ONE launch (grit_transpose_bf16_grouped, grit_amd/csrc/transpose.hip).

The input gradient of a Swin Mlp's fc2 goes through the fused GELU' GEMM, whose B operand is K-contiguous: it needs
fc2.weight^T.  As `w.t().contiguous()` inside each block's backward that is 24 dependent ~10 us launches per training step; the
weights only change at the optimizer step, so SwinTransformer.forward refreshes all of them at once (`refresh`) and the backward
nodes pick theirs up (`lookup`) -- or fall back to the transpose of their own when the copy is missing or stale.

A copy is valid for one (grit_amd.ops.weights_epoch, data_ptr, tensor._version): the flat optimizer rewrites the compute weights
through a raw kernel launch, which no version counter sees -- that is what weights_epoch is for.  `lookup` checks the tag, so a weight
changed in any visible way between forward and backward is transposed again rather than used stale."""
import ctypes
import os

import torch

from grit_amd import lib as _lib
from grit_amd.ops import backend, weights_epoch

ENABLED = os.environ.get("GRIT_TRANSPOSED_WEIGHTS", "1") != "0"  # A/B knob: 0 = every backward node transposes its own weight


def _tag(w):
    return (weights_epoch.current(), w.data_ptr(), w._version)


def refresh(weights):
    """Make / renew the transposed copies of the given 2-D bf16 weights (rows, cols multiples of 64) in one launch per 32 of them."""
    if not ENABLED:
        return
    stale = []
    for w in weights:
        if not (w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 2 and w.is_contiguous() and w.shape[0] % 64 == 0
                and w.shape[1] % 64 == 0 and w.data_ptr() % 16 == 0):
            continue
        hit = getattr(w, "_grit_transposed", None)
        # inside the capture of a whole training step the copies are part of the graph: every replay transposes the weights as the
        # optimizer step of the previous replay left them, whatever the host-side tags say
        if hit is None or hit[0] != _tag(w) or backend.train_capture():
            stale.append(w)
    if not stale or backend.foreign_capture():
        return
    lib = _lib.load()
    with torch.no_grad():
        for i in range(0, len(stale), _lib.TRANSPOSE_GROUP_MAX):
            chunk = stale[i:i + _lib.TRANSPOSE_GROUP_MAX]
            table = (_lib.TransposeJob * len(chunk))()
            outs = []
            for t, w in enumerate(chunk):
                hit = getattr(w, "_grit_transposed", None)
                out = hit[1] if (hit is not None and hit[1].shape == (w.shape[1], w.shape[0]) and hit[1].device == w.device) \
                    else torch.empty((w.shape[1], w.shape[0]), dtype=w.dtype, device=w.device)
                outs.append(out)
                table[t] = _lib.TransposeJob(w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1])
            with _lib.device_guard(chunk[0].device):
                _lib.check(lib.grit_transpose_bf16_grouped(table, len(chunk), _lib.current_stream_ptr()), "grit_transpose_bf16_grouped")
            for w, out in zip(chunk, outs):
                w._grit_transposed = (_tag(w), out)


def lookup(w):
    """The transposed copy of `w` if one exists for its current value, else None."""
    hit = getattr(w, "_grit_transposed", None) if ENABLED else None
    return hit[1] if (hit is not None and hit[0] == _tag(w)) else None


def refresh_linears(module):
    """refresh() for the weights of every nn.Linear inside `module` (the decoders and the grid net: the input gradients of their short
    maps run as NT products on these copies, grit_amd/ops/gemm.py prefers_own_short).  The list of Linear modules is kept on `module`."""
    if not ENABLED:
        return
    mods = getattr(module, "_grit_linear_modules", None)
    if mods is None:
        mods = [m for m in module.modules() if isinstance(m, torch.nn.Linear)]
        object.__setattr__(module, "_grit_linear_modules", mods)
    ws = [m.weight for m in mods if m.weight.requires_grad]
    attn = getattr(module, "_grit_mha_modules", None)
    if attn is None:
        attn = [m for m in module.modules() if isinstance(m, torch.nn.MultiheadAttention) and m.in_proj_weight is not None]
        object.__setattr__(module, "_grit_mha_modules", attn)
    ws += [m.in_proj_weight for m in attn if m.in_proj_weight.requires_grad]  # (the packed in-projection node of ops/linear.py)
    if ws and ws[0].is_cuda and ws[0].dtype == torch.bfloat16:
        refresh(ws)
