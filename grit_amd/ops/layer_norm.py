"""LayerNorm over the last dimension on the gfx950 streaming kernels (grit_layernorm_{fwd,bwd}).

Used by the Swin backbone (norm1 / norm2 / patch-merging / patch-embed norms): there LayerNorm is a pure HBM
stream over maps of up to 819 200 tokens and torch's bf16 kernels run at a fraction of the bandwidth.  A device tensor
whose channel count / dtype the kernels do not cover RAISES (no silent library fallback on the hot path)."""
import ctypes

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend
from grit_amd.ops.linear import (SlabGroup, _own_input_grad, _own_linear, defer_slab_group, defer_weight_bias_grad, finish_group, fork, join, on_stream, single_use_now,
                                 slab_sum)
from grit_amd.ops.profiling import gemm_work, timed

SUPPORTED_C = (128, 256, 512, 1024, 2048, 4096)
LN_BWD_PARTIALS = 1024  # GRIT_LN_BWD_PARTIALS in include/grit_hip.h


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class _LayerNormFn(Function):

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        xb, wb = int(x2.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        with _lib.device_guard(x.device):
            st = _lib.load().grit_layernorm_fwd(_ptr(x2), _ptr(weight), _ptr(bias), rows, C, eps, xb, wb, _ptr(y), _ptr(mean),
                                                _ptr(rstd), _lib.current_stream_ptr())
        _lib.check(st, "grit_layernorm_fwd")
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        dy2 = dy.reshape(rows, C)
        if not dy2.is_contiguous() or dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype).contiguous()
        dx = torch.empty_like(x2)
        # per-workgroup partial sums; the kernel launches min(ceil(rows / rows_per_block), LN_BWD_PARTIALS) workgroups and every
        # one of them writes its row, so the buffer needs no zero fill
        rows_per_block = 4 * (64 // min(C // 8, 64))
        nblk = min(-(-rows // rows_per_block), LN_BWD_PARTIALS)
        base = torch.empty(2, LN_BWD_PARTIALS, C, dtype=torch.float32, device=x2.device)
        xb, wb = int(x2.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        with _lib.device_guard(x2.device):
            st = _lib.load().grit_layernorm_bwd(_ptr(x2), _ptr(weight), _ptr(dy2), _ptr(mean), _ptr(rstd), rows, C, xb, wb,
                                                _ptr(dx), _ptr(base[0]), _ptr(base[1]), _lib.current_stream_ptr())
        _lib.check(st, "grit_layernorm_bwd")
        sums = slab_sum(base, weight.dtype, slabs=nblk)  # [2, C]: dgamma, dbeta
        return dx.view(ctx.shape), sums[0], sums[1], None


def layer_norm(x, weight, bias, eps=1e-5):
    ov = backend.override()
    if ov is not None and hasattr(ov, "layer_norm"):
        return ov.layer_norm(x, weight, bias, eps)
    C = x.shape[-1]
    fits = (x.is_cuda and C in SUPPORTED_C and weight is not None and bias is not None
            and x.dtype in (torch.float32, torch.bfloat16) and weight.dtype == bias.dtype
            and (weight.dtype == x.dtype or (x.dtype == torch.bfloat16 and weight.dtype == torch.float32)))
    if not fits:
        if x.is_cuda:  # no silent library fallback on the device: a shape / dtype outside the kernels' range is a loud error
            raise _lib.GritHipError(
                "layer_norm: no HIP kernel for C = %d (supported: %s), x %s, weight %s, bias %s -- every LayerNorm of GRIT fits; "
                "extend grit_layernorm_* or call torch.nn.functional.layer_norm explicitly" %
                (C, SUPPORTED_C, x.dtype, None if weight is None else weight.dtype, None if bias is None else bias.dtype))
        return F.layer_norm(x, (C,), weight, bias, eps)  # host tensors (config 1 plumbing on CPU): the library op
    return _LayerNormFn.apply(x, weight.contiguous(), bias.contiguous(), float(eps))


class _MergeLayerNormFn(Function):
    """LayerNorm(4C) of the 2 x 2 patch-merged view of a token map [B, H, W, C] (H, W even) without materialising the view
    (grit_merge_layernorm_{fwd,bwd}; reference models/common/swin_model.py:279-288)."""

    @staticmethod
    def forward(ctx, x, H, W, weight, bias, eps):
        B, C = x.shape[0], x.shape[-1]
        x4 = x.reshape(B, H, W, C)
        if not x4.is_contiguous():
            x4 = x4.contiguous()
        rows = B * (H // 2) * (W // 2)
        y = torch.empty((rows, 4 * C), dtype=x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        xb, wb = int(x.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        with _lib.device_guard(x.device):
            st = _lib.load().grit_merge_layernorm_fwd(_ptr(x4), B, H, W, C, _ptr(weight), _ptr(bias), eps, xb, wb, _ptr(y), _ptr(mean),
                                                      _ptr(rstd), _lib.current_stream_ptr())
        _lib.check(st, "grit_merge_layernorm_fwd")
        ctx.save_for_backward(x4, weight, mean, rstd)
        ctx.shape = x.shape
        return y.view(B, (H // 2) * (W // 2), 4 * C)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x4, weight, mean, rstd = ctx.saved_tensors
        B, H, W, C = x4.shape
        rows = B * (H // 2) * (W // 2)
        dy2 = dy.reshape(rows, 4 * C)
        if not dy2.is_contiguous() or dy2.dtype != x4.dtype:
            dy2 = dy2.to(x4.dtype).contiguous()
        dx = torch.empty_like(x4)
        rows_per_block = 4 * (64 // min(4 * C // 8, 64))
        nblk = min(-(-rows // rows_per_block), LN_BWD_PARTIALS)
        base = torch.empty(2, LN_BWD_PARTIALS, 4 * C, dtype=torch.float32, device=x4.device)
        xb, wb = int(x4.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        with _lib.device_guard(x4.device):
            st = _lib.load().grit_merge_layernorm_bwd(_ptr(x4), B, H, W, C, _ptr(weight), _ptr(dy2), _ptr(mean), _ptr(rstd), xb, wb,
                                                      _ptr(dx), _ptr(base[0]), _ptr(base[1]), _lib.current_stream_ptr())
        _lib.check(st, "grit_merge_layernorm_bwd")
        sums = slab_sum(base, weight.dtype, slabs=nblk)
        return dx.view(ctx.shape), None, None, sums[0], sums[1], None


def merge_layer_norm(x, H, W, weight, bias, eps=1e-5):
    """LayerNorm(4C)(patch_merge(x)) for x [B, H*W, C] (or [B, H, W, C]), or None when the fused form does not apply (odd H / W, a
    channel count or dtype outside the kernels, an injected backend): the caller then builds the merged view itself."""
    if backend.override() is not None:
        return None
    C = x.shape[-1]
    fits = (x.is_cuda and H % 2 == 0 and W % 2 == 0 and C % 8 == 0 and 4 * C in SUPPORTED_C and weight is not None and bias is not None
            and x.dtype in (torch.float32, torch.bfloat16) and weight.dtype == bias.dtype
            and (weight.dtype == x.dtype or (x.dtype == torch.bfloat16 and weight.dtype == torch.float32)))
    if not fits:
        return None
    return _MergeLayerNormFn.apply(x, H, W, weight.contiguous(), bias.contiguous(), float(eps))


class _AddLayerNormFn(Function):
    """(shortcut, branch, scale) -> (x, LayerNorm(x)) with x = shortcut + scale * branch; scale [B] f32 or None."""

    @staticmethod
    def forward(ctx, shortcut, branch, scale, weight, bias, eps):
        C = shortcut.shape[-1]
        s2, b2 = shortcut.reshape(-1, C), branch.reshape(-1, C)
        s2 = s2 if s2.is_contiguous() else s2.contiguous()
        b2 = b2 if b2.is_contiguous() else b2.contiguous()
        rows = s2.shape[0]
        x = torch.empty_like(s2)
        y = torch.empty_like(s2)
        mean = torch.empty(rows, dtype=torch.float32, device=s2.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=s2.device)
        xb, wb = int(s2.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        per_sample = rows // shortcut.shape[0]
        with _lib.device_guard(s2.device):
            st = _lib.load().grit_add_layernorm_fwd(_ptr(s2), _ptr(b2), _ptr(scale) if scale is not None else None, per_sample,
                                                    0.0, None, _ptr(weight), _ptr(bias), rows, C, eps, xb, wb, _ptr(x), _ptr(y),
                                                    _ptr(mean), _ptr(rstd), _lib.current_stream_ptr())
        _lib.check(st, "grit_add_layernorm_fwd")
        ctx.save_for_backward(x, weight, mean, rstd, scale)
        ctx.shape = shortcut.shape
        ctx.set_materialize_grads(False)  # an unused output (the post-norm decoders drop x) arrives as None, not a zero map
        return x.view(shortcut.shape), y.view(shortcut.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gy):
        x2, weight, mean, rstd, scale = ctx.saved_tensors
        dx, d_branch, sums = _add_layer_norm_backward(x2, weight, mean, rstd, scale, gx, gy, ctx.shape[0], False, 0.0, None)
        return dx.view(ctx.shape), d_branch.view(ctx.shape), None, sums[0], sums[1], None


def _add_layer_norm_backward(x2, weight, mean, rstd, scale, gx, gy, batch, branch_colsum, drop_p, seed_dev, group=None):
    """(dx, d_branch, sums): sums[0] = dgamma, sums[1] = dbeta, and with branch_colsum sums[2] = column sums of d_branch
    (the bias gradient of the Linear that produced the branch), all from the one grit_add_layernorm_bwd launch.  With `group`
    (a SlabGroup) the per-workgroup partials are summed by the group's launch: `sums` is filled by group.run()."""
    rows, C = x2.shape
    if gy is None:
        gy = torch.zeros_like(x2)
    gy2 = gy.reshape(rows, C)
    if not gy2.is_contiguous() or gy2.dtype != x2.dtype:
        gy2 = gy2.to(x2.dtype).contiguous()
    gx2 = None
    if gx is not None:
        gx2 = gx.reshape(rows, C)
        if not gx2.is_contiguous() or gx2.dtype != x2.dtype:
            gx2 = gx2.to(x2.dtype).contiguous()
    dx = torch.empty_like(x2)
    own_branch = scale is not None or drop_p > 0
    d_branch = torch.empty_like(x2) if own_branch else dx
    rows_per_block = 4 * (64 // min(C // 8, 64))
    nblk = min(-(-rows // rows_per_block), LN_BWD_PARTIALS)
    n_sums = 3 if branch_colsum else 2
    base = torch.empty(n_sums, LN_BWD_PARTIALS, C, dtype=torch.float32, device=x2.device)
    xb, wb = int(x2.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
    with _lib.device_guard(x2.device):
        st = _lib.load().grit_add_layernorm_bwd(
            _ptr(x2), _ptr(weight), _ptr(gy2), _ptr(gx2) if gx2 is not None else None, _ptr(mean), _ptr(rstd),
            _ptr(scale) if scale is not None else None, rows // batch, float(drop_p),
            _ptr(seed_dev) if drop_p > 0 else None, rows, C, xb, wb, _ptr(dx), _ptr(d_branch) if own_branch else None,
            _ptr(base[0]), _ptr(base[1]), _ptr(base[2]) if branch_colsum else None, _lib.current_stream_ptr())
    _lib.check(st, "grit_add_layernorm_bwd")
    if group is not None:
        return dx, d_branch, group.add(base, weight.dtype, slabs=nblk)
    return dx, d_branch, slab_sum(base, weight.dtype, slabs=nblk)


def _residual_linear(inp, lin_w, lin_b, s2, scale, rows_per_sample):
    """s2 + scale[sample] * (inp @ lin_w^T + lin_b) as ONE launch of the own GEMM (grit_amd/ops/gemm.py long_linear_residual), or None
    where that does not apply (short maps, fp32, shapes the policy leaves to the library)."""
    if not (inp.is_cuda and inp.dtype == torch.bfloat16 and s2.dtype == torch.bfloat16 and inp.numel() // inp.shape[-1] >= 8192):
        return None
    from grit_amd.ops import gemm as _gemm
    x2 = inp.reshape(-1, inp.shape[-1])
    return _gemm.long_linear_residual(x2 if x2.is_contiguous() else x2.contiguous(), lin_w, lin_b, s2, scale, rows_per_sample)


def _projection_input_grad(ctx, d_branch, inp, lin_w):
    """d(inp) of the projection inside _LinearAddLayerNormFn: d_branch @ W on the own kernels where the policies take the shape; with the
    backward of ReLU + dropout in the GEMM's epilogue when the node took that over from linear_relu_dropout (ctx.drelu)."""
    h = ctx.drelu
    if h is not None:
        from grit_amd.ops import gemm as _gemm
        from grit_amd.ops import transposed
        from grit_amd.ops.glue import relu_dropout_backward
        inp2 = inp.reshape(-1, inp.shape[-1])
        wt = transposed.lookup(ctx.lin_w_obj)
        if wt is not None and _gemm.supported(d_branch, wt) and inp2.data_ptr() % 16 == 0:
            return _gemm.gemm_nt_relu(d_branch, wt, _gemm.DRELU, aux=inp2, p=h["p"]).view(inp.shape)
        with timed("gemm_lib", **gemm_work(d_branch.shape[0], lin_w.shape[1], lin_w.shape[0])):
            d = torch.mm(d_branch, lin_w)  # (no transposed copy of the weight at hand: the library's product, then the mask -- the contract holds)
        return relu_dropout_backward(inp2, d, h["p"], h["seed"]).view(inp.shape)
    d_inp = _own_input_grad(d_branch, ctx.lin_w_obj, inp.shape)
    if d_inp is None:
        with timed("gemm_lib", **gemm_work(d_branch.shape[0], lin_w.shape[1], lin_w.shape[0])):
            d_inp = torch.mm(d_branch, lin_w).view(inp.shape)
    return d_inp


class _LinearAddLayerNormFn(Function):
    """(inp, W, b, shortcut, scale) -> (x, LayerNorm(x)) with x = shortcut + scale * (inp @ W^T + b): the output projection
    of a Swin branch (attn.proj / mlp.fc2), the residual connection and the LayerNorm that follows, as one node -- so the
    LayerNorm backward kernel, which already streams the branch gradient, also delivers the Linear's bias gradient."""

    @staticmethod
    def forward(ctx, inp, lin_w, lin_b, shortcut, scale, weight, bias, eps, drop_p, seed_dev, single_use=False):
        ctx.single_use = single_use
        ctx.sum_params = (weight, bias, lin_b) if single_use else None  # the parameters whose gradients the node's sums are
        ctx.lin_w_param = lin_w if single_use else None
        ctx.lin_w_obj = lin_w  # the tensor object of the call: transposed copies are attached to it (grit_amd/ops/transposed.py)
        C = shortcut.shape[-1]
        s2 = shortcut.reshape(-1, C)
        s2 = s2 if s2.is_contiguous() else s2.contiguous()
        rows = s2.shape[0]
        x = _residual_linear(inp, lin_w, lin_b, s2, scale, rows // shortcut.shape[0]) if drop_p == 0 else None
        y = torch.empty_like(s2)
        mean = torch.empty(rows, dtype=torch.float32, device=s2.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=s2.device)
        xb, wb = int(s2.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        if x is not None:  # long maps: the projection wrote x = shortcut + factor * branch itself; the LayerNorm reads x only
            with _lib.device_guard(s2.device):
                st = _lib.load().grit_layernorm_fwd(_ptr(x), _ptr(weight), _ptr(bias), rows, C, eps, xb, wb, _ptr(y), _ptr(mean), _ptr(rstd),
                                                    _lib.current_stream_ptr())
            _lib.check(st, "grit_layernorm_fwd")
        else:
            branch = _own_linear(inp, lin_w, lin_b)  # long maps: the own four-wave kernel where it is the faster one
            if branch is None:
                with timed("gemm_lib", **gemm_work(inp.numel() // inp.shape[-1], lin_w.shape[0], lin_w.shape[1])):
                    branch = F.linear(inp, lin_w, lin_b)
            b2 = branch.reshape(-1, C)
            x = torch.empty_like(s2)
            with _lib.device_guard(s2.device):
                st = _lib.load().grit_add_layernorm_fwd(_ptr(s2), _ptr(b2), _ptr(scale) if scale is not None else None,
                                                        rows // shortcut.shape[0], float(drop_p),
                                                        _ptr(seed_dev) if drop_p > 0 else None, _ptr(weight), _ptr(bias), rows, C,
                                                        eps, xb, wb, _ptr(x), _ptr(y), _ptr(mean), _ptr(rstd),
                                                        _lib.current_stream_ptr())
            _lib.check(st, "grit_add_layernorm_fwd")
        ctx.save_for_backward(x, weight, mean, rstd, scale, inp, lin_w, seed_dev)
        ctx.shape, ctx.drop_p = shortcut.shape, drop_p
        # `inp` = dropout(relu(.)) from linear_relu_dropout (grit_amd/ops/linear.py): this node's input-gradient GEMM applies the backward
        # of ReLU + dropout in its epilogue, and says so in the holder -- the producer's backward then takes the gradient as it comes
        ctx.drelu = None
        holder = getattr(inp, "_grit_relu_holder", None)
        if holder is not None and ctx.needs_input_grad[0] and not holder["fused"]:
            from grit_amd.ops import gemm as _gemm
            if (inp.dtype == torch.bfloat16 and inp.is_cuda and inp.is_contiguous()
                    and _gemm.prefers_own_short(rows, inp.shape[-1], lin_w.shape[0])):
                holder["fused"] = True
                ctx.drelu = holder
        ctx.set_materialize_grads(False)  # see _AddLayerNormFn
        return x.view(shortcut.shape), y.view(shortcut.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gy):
        from grit_amd.ops.linear import weight_grad
        x2, weight, mean, rstd, scale, inp, lin_w, seed_dev = ctx.saved_tensors
        group = SlabGroup()  # LayerNorm sums + the projection's weight gradient: one reduction launch for the node
        dx, d_branch, sums = _add_layer_norm_backward(x2, weight, mean, rstd, scale, gx, gy, ctx.shape[0], True, ctx.drop_p,
                                                      seed_dev, group)
        inp2 = inp.reshape(-1, inp.shape[-1])
        if not inp2.is_contiguous():
            inp2 = inp2.contiguous()
        d_inp = d_lin_w = None
        # drop path: d_branch = scale[b] * dx is exactly zero in the rows of dropped samples -- the projection's weight gradient skips them
        rs = (scale, d_branch.shape[0] // ctx.shape[0]) if (scale is not None and ctx.drop_p == 0) else None
        deferred = defer_weight_bias_grad(d_branch, inp2, lin_w, None, ctx.needs_input_grad[1], False, ctx.single_use, row_scale=rs)
        if deferred is not None:  # short map: the projection's weight gradient joins the scope's grouped launch
            esz = sums.element_size() * sums.shape[-1]
            if not (ctx.sum_params is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[5] and ctx.needs_input_grad[6]
                    and sums.dtype == lin_w.dtype
                    and defer_slab_group(group, [(ctx.sum_params[0], sums.data_ptr()), (ctx.sum_params[1], sums.data_ptr() + esz),
                                                 (ctx.sum_params[2], sums.data_ptr() + 2 * esz)])):
                group.run()  # ... and so do the node's LayerNorm / bias sums when all three gradients are wanted
            if ctx.needs_input_grad[0]:
                d_inp = _projection_input_grad(ctx, d_branch, inp, lin_w)
            return (d_inp, deferred[0], sums[2].to(lin_w.dtype), dx.view(ctx.shape), None, sums[0], sums[1], None, None, None,
                    None)
        # small maps inside a deferral scope: the projection's weight gradient beside the chain (grit_amd/ops/linear.py fork)
        side = fork(d_branch, inp2, rows=d_branch.shape[0], single_use=ctx.single_use) \
            if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1]) else None
        if side is not None and getattr(side, "deferred", False):
            group.run()  # the LayerNorm sums stay on this stream; the weight gradient of a small map has no partials (S = 1)
            with on_stream(side):
                d_lin_w = weight_grad(d_branch, inp2)
        if ctx.needs_input_grad[0]:
            d_inp = _projection_input_grad(ctx, d_branch, inp, lin_w)
        if d_lin_w is None:
            d_lin_w = weight_grad(d_branch, inp2, group, param=lin_w, row_scale=rs) if ctx.needs_input_grad[1] else None
            sp = ctx.sum_params
            # long maps: the node's reductions (weight-gradient partials, LayerNorm / bias sums) join the scope's grouped launch
            finish_group(group, side is None and sp is not None and sums.dtype == lin_w.dtype and ctx.needs_input_grad[2]
                         and ctx.needs_input_grad[5] and ctx.needs_input_grad[6],
                         [] if sp is None else [(ctx.lin_w_param, d_lin_w), (sp[0], sums[0]), (sp[1], sums[1]), (sp[2], sums[2])])
        join(side, d_lin_w)
        return d_inp, d_lin_w, sums[2].to(lin_w.dtype), dx.view(ctx.shape), None, sums[0], sums[1], None, None, None, None


MIN_ROWS_FUSED = 512  # below this torch's own chain is as good


def linear_add_layer_norm(inp, linear, shortcut, scale, weight, bias, eps=1e-5, dropout_p=0.0, training=False, single_use=False):
    """x = shortcut + scale[b] * dropout(linear(inp));  returns (x, layer_norm(x)).  `linear`: an nn.Linear-like module
    with bias; scale: per-sample stochastic-depth factors or None; dropout_p applies (in training) between the projection
    and the residual add, as nn.Dropout does in the post-norm decoder layers."""
    p = float(dropout_p) if training else 0.0
    C = shortcut.shape[-1]
    fits = (backend.override() is None and shortcut.is_cuda and torch.is_grad_enabled() and C in SUPPORTED_C and C <= 1024
            and linear.bias is not None and linear.weight.shape[0] == C and inp.dtype == shortcut.dtype == linear.weight.dtype
            and shortcut.dtype in (torch.float32, torch.bfloat16) and weight is not None and bias is not None
            and weight.dtype == bias.dtype and weight.dtype == linear.weight.dtype and inp.shape[:-1] == shortcut.shape[:-1]
            and (inp.requires_grad or linear.weight.requires_grad) and not torch.is_autocast_enabled()
            and shortcut.numel() // C >= MIN_ROWS_FUSED and 0.0 <= p < 1.0)
    if not fits:
        if (p == 0 and backend.override() is None and shortcut.is_cuda and shortcut.dtype == torch.bfloat16
                and C in SUPPORTED_C and linear.bias is not None and weight is not None and bias is not None
                and weight.dtype == bias.dtype == torch.bfloat16 and not torch.is_autocast_enabled()
                and not (torch.is_grad_enabled() and (inp.requires_grad or shortcut.requires_grad or linear.weight.requires_grad))):
            # inference / a frozen stage: the projection writes x = shortcut + branch itself (same launch as in the training node),
            # the LayerNorm reads x only
            s2 = shortcut.reshape(-1, C)
            x = _residual_linear(inp, linear.weight, linear.bias, s2 if s2.is_contiguous() else s2.contiguous(),
                                 None if scale is None else scale.reshape(-1).float().contiguous(), s2.shape[0] // shortcut.shape[0])
            if x is not None:
                return x.view(shortcut.shape), layer_norm(x.view(shortcut.shape), weight, bias, eps)
        branch = linear(inp)
        if p > 0:
            branch = F.dropout(branch, p, True)
        return add_layer_norm(shortcut, branch, scale, weight, bias, eps)
    if scale is not None:
        scale = scale.reshape(-1).float().contiguous()
    seed_dev = backend.dropout_seed(inp.device) if p > 0 else None
    return _LinearAddLayerNormFn.apply(inp, linear.weight, linear.bias, shortcut, scale, weight.contiguous(), bias.contiguous(),
                                       float(eps), p, seed_dev, single_use_now(single_use or getattr(linear, 'single_use', False)))


def add_layer_norm(shortcut, branch, scale, weight, bias, eps=1e-5):
    """x = shortcut + scale[b] * branch (scale None: plain add); returns (x, layer_norm(x)) from one pass over HBM when
    the streaming kernel covers the shape, otherwise the unfused composition (same values either way)."""
    C = shortcut.shape[-1]
    fits = (backend.override() is None and shortcut.is_cuda and C in SUPPORTED_C and shortcut.shape == branch.shape
            and shortcut.dtype == branch.dtype and shortcut.dtype in (torch.float32, torch.bfloat16)
            and weight is not None and bias is not None and weight.dtype == bias.dtype and shortcut.dim() >= 2
            and (weight.dtype == shortcut.dtype or (shortcut.dtype == torch.bfloat16 and weight.dtype == torch.float32))
            and not torch.is_autocast_enabled())
    if not fits:
        if scale is None:
            x = shortcut + branch
        else:
            x = torch.addcmul(shortcut, branch, scale.view(-1, *([1] * (shortcut.dim() - 1))).to(shortcut.dtype))
        return x, layer_norm(x, weight, bias, eps)
    if scale is not None:
        scale = scale.reshape(-1).float().contiguous()
    return _AddLayerNormFn.apply(shortcut, branch, scale, weight.contiguous(), bias.contiguous(), float(eps))


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict keys) whose forward runs the streaming HIP kernels."""

    def forward(self, input):
        if len(self.normalized_shape) != 1 or not self.elementwise_affine:
            return super().forward(input)
        return layer_norm(input, self.weight, self.bias, self.eps)
