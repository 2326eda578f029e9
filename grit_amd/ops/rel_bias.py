"""Relative-position bias lookup of Swin window attention on the HIP kernels (grit_relbias_{fwd,bwd}).

    bias[h, i, j] = table[index[i, j], h]       (reference models/common/swin_model.py:168-171)

Forward is one small gather kernel writing the [nH, N, N] fp32 slab the attention kernels read; backward sums d(bias)
over the positions that share a table row, from a host-sorted position list (no atomics, no sort per step)."""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend

import os

# GRIT_GROUPED_REL_BIAS_BWD (default 1): the grouped gather is one autograd node whose backward is one launch for all tables; 0: a node per module
GROUPED_BACKWARD = os.environ.get("GRIT_GROUPED_REL_BIAS_BWD", "1") != "0"

_SORTED = {}  # (device, n_rows, index data_ptr, numel) -> (order int32 [n_pos], offsets int32 [n_rows + 1])


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _sorted_positions(index, n_rows):
    key = (str(index.device), n_rows, index.data_ptr(), index.numel())
    hit = _SORTED.get(key)
    if hit is None:
        flat = index.detach().reshape(-1).cpu()  # one-time host round trip per module (the index never changes)
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= n_rows):
            raise RuntimeError("relative_position_index out of range of the bias table")
        order = torch.argsort(flat, stable=True).to(torch.int32)
        counts = torch.bincount(flat, minlength=n_rows)
        offsets = torch.cat((counts.new_zeros(1), counts.cumsum(0))).to(torch.int32)
        hit = (order.to(index.device), offsets.to(index.device))
        if len(_SORTED) > 256:
            _SORTED.clear()
        _SORTED[key] = hit
    return hit


def _table_grad(index, meta, dbias):
    n_rows, nH, n_pos, dtype = meta
    order, offsets = _sorted_positions(index, n_rows)
    dbias = dbias.float().contiguous()
    dtable = torch.empty(n_rows, nH, dtype=dtype, device=dbias.device)
    with _lib.device_guard(dbias.device):
        st = _lib.load().grit_relbias_bwd(_ptr(dbias), _ptr(order), _ptr(offsets), n_rows, nH, n_pos,
                                          int(dtype == torch.bfloat16), _ptr(dtable), _lib.current_stream_ptr())
    _lib.check(st, "grit_relbias_bwd")
    return dtable


class _RelBiasFn(Function):

    @staticmethod
    def forward(ctx, table, index):
        n_rows, nH = table.shape
        n_pos = index.numel()
        out = torch.empty((nH,) + tuple(index.shape), dtype=torch.float32, device=table.device)
        with _lib.device_guard(table.device):
            st = _lib.load().grit_relbias_fwd(_ptr(table), _ptr(index), n_rows, nH, n_pos, int(table.dtype == torch.bfloat16),
                                              _ptr(out), _lib.current_stream_ptr())
        _lib.check(st, "grit_relbias_fwd")
        ctx.index, ctx.meta = index, (n_rows, nH, n_pos, table.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dbias):
        return _table_grad(ctx.index, ctx.meta, dbias), None


class _RelBiasGivenFn(Function):
    """_RelBiasFn whose forward result already exists (relative_position_bias_grouped computed it with the other modules'): no launch
    forward, the same backward."""

    @staticmethod
    def forward(ctx, table, index, given):
        ctx.index, ctx.meta = index, (table.shape[0], table.shape[1], index.numel(), table.dtype)
        return given.view(given.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dbias):
        return _table_grad(ctx.index, ctx.meta, dbias), None, None


class _RelBiasGroupedFn(Function):
    """The gathers of n modules as ONE node: one launch forward (grit_relbias_fwd_grouped) and -- when the gradients of all its outputs have
    arrived, i.e. behind the first block's attention backward -- one launch backward (grit_relbias_bwd_grouped) instead of one per module."""

    @staticmethod
    def forward(ctx, n, *args):
        indices, tables = args[:n], args[n:]
        outs = [torch.empty((t.shape[1],) + tuple(i.shape), dtype=torch.float32, device=t.device) for t, i in zip(tables, indices)]
        table = (_lib.RelbiasJob * n)()
        for k, (t, i, o) in enumerate(zip(tables, indices, outs)):
            table[k] = _lib.RelbiasJob(t.data_ptr(), i.data_ptr(), o.data_ptr(), t.shape[0], t.shape[1], i.numel(), int(t.dtype == torch.bfloat16))
        with _lib.device_guard(tables[0].device):
            st = _lib.load().grit_relbias_fwd_grouped(table, n, _lib.current_stream_ptr())
        _lib.check(st, "grit_relbias_fwd_grouped")
        ctx.indices = indices
        ctx.meta = [(t.shape[0], t.shape[1], i.numel(), t.dtype) for t, i in zip(tables, indices)]
        ctx.set_materialize_grads(False)
        # (a frozen table's slab must not carry a gradient requirement into its block: the attention of a frozen stage would run a backward)
        ctx.mark_non_differentiable(*[o for o, t in zip(outs, tables) if not t.requires_grad])
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dbias):
        n = len(ctx.meta)
        grads, jobs, keep = [None] * n, [], []
        for k, (d, idx, (n_rows, nH, n_pos, dtype)) in enumerate(zip(dbias, ctx.indices, ctx.meta)):
            if d is None or not ctx.needs_input_grad[1 + n + k]:
                continue
            order, offsets = _sorted_positions(idx, n_rows)
            d = d if (d.dtype == torch.float32 and d.is_contiguous()) else d.float().contiguous()
            grads[k] = torch.empty(n_rows, nH, dtype=dtype, device=d.device)
            keep.append(d)
            jobs.append(_lib.RelbiasBwdJob(d.data_ptr(), order.data_ptr(), offsets.data_ptr(), grads[k].data_ptr(), n_rows, nH, n_pos,
                                           int(dtype == torch.bfloat16)))
        if jobs:
            table = (_lib.RelbiasBwdJob * len(jobs))(*jobs)
            with _lib.device_guard(keep[0].device):
                st = _lib.load().grit_relbias_bwd_grouped(table, len(jobs), _lib.current_stream_ptr())
            _lib.check(st, "grit_relbias_bwd_grouped")
        return (None,) * (1 + n) + tuple(grads)


def relative_position_bias_grouped(tables, indices):
    """[relative_position_bias(t, i) for t, i in zip(tables, indices)] from ONE launch (grit_relbias_fwd_grouped): float32 tensors
    [nH, N, N].  With autograd on and a table that wants a gradient they are the outputs of ONE node (its backward is one launch too);
    otherwise plain tensors (views of one buffer).  The caller hands each to `relative_position_bias(.., given=)`.  None where the kernel
    path does not apply."""
    ok = (backend.override() is None and len(tables) > 0 and len(tables) <= _lib.RELBIAS_GROUP_MAX
          and all(t.is_cuda and i.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and i.dtype == torch.int64 and t.is_contiguous()
                  and i.is_contiguous() and t.dim() == 2 for t, i in zip(tables, indices)))
    if not ok:
        return None
    if GROUPED_BACKWARD and torch.is_grad_enabled() and any(t.requires_grad for t in tables):
        for t, i in zip(tables, indices):
            if t.requires_grad:
                _sorted_positions(i, t.shape[0])  # built outside the autograd thread, before any graph capture
        return list(_RelBiasGroupedFn.apply(len(tables), *indices, *tables))
    sizes = [t.shape[1] * i.numel() for t, i in zip(tables, indices)]
    with torch.no_grad():
        buf = torch.empty(sum(sizes), dtype=torch.float32, device=tables[0].device)
        outs, off = [], 0
        table = (_lib.RelbiasJob * len(tables))()
        for k, (t, i, n) in enumerate(zip(tables, indices, sizes)):
            out = buf[off:off + n].view((t.shape[1],) + tuple(i.shape))
            off += n
            outs.append(out)
            table[k] = _lib.RelbiasJob(t.data_ptr(), i.data_ptr(), out.data_ptr(), t.shape[0], t.shape[1], i.numel(),
                                       int(t.dtype == torch.bfloat16))
        with _lib.device_guard(tables[0].device):
            st = _lib.load().grit_relbias_fwd_grouped(table, len(tables), _lib.current_stream_ptr())
        _lib.check(st, "grit_relbias_fwd_grouped")
    return outs


def relative_position_bias(table, index, given=None):
    """table [n_rows, nH] (f32 / bf16), index [N, N] int64 -> [nH, N, N] float32.  given: the values, already gathered by
    relative_position_bias_grouped for this table's CURRENT contents (the caller's promise) -- only the autograd node is built."""
    ov = backend.override()
    fits = (ov is None and table.is_cuda and index.is_cuda and table.dtype in (torch.float32, torch.bfloat16)
            and index.dtype == torch.int64 and table.is_contiguous() and index.is_contiguous())
    if not fits:
        if not table.is_cuda and ov is None:
            raise _lib.GritHipError("Not implemented on the CPU: relative_position_bias runs on the HIP kernels "
                                    "(tests inject oracle ops with grit_amd.ops.backend.use_reference_ops)")
        n = index.shape[0]
        return table[index.reshape(-1)].view(n, index.shape[1], -1).permute(2, 0, 1).contiguous().float()
    if torch.is_grad_enabled() and table.requires_grad:
        _sorted_positions(index, table.shape[0])  # built outside the autograd thread, before any graph capture
    if given is not None and given.shape == (table.shape[1],) + tuple(index.shape) and given.device == table.device:
        if given.requires_grad:
            return given  # an output of the grouped node: its gradient flows there
        return _RelBiasGivenFn.apply(table, index, given) if (torch.is_grad_enabled() and table.requires_grad) else given
    return _RelBiasFn.apply(table, index)
