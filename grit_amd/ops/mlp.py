"""The Mlp of a Swin block (fc1 -> exact GELU -> fc2, reference models/common/swin_model.py:31-37) and the block tail that
follows it (drop-path, residual add, the next LayerNorm; :289-298) as ONE autograd node on the fused-epilogue GEMM.

  forward   fc1 + bias + GELU            grit_gemm_bf16_nt / GRIT_GEMM_BIAS_GELU   (pre-activation and activation in one pass)
            fc2 + bias                   library GEMM
            residual + LayerNorm         grit_add_layernorm_fwd
  backward  LayerNorm / residual         grit_add_layernorm_bwd   (also fc2's bias gradient)
            fc2 input gradient x GELU'   grit_gemm_bf16_nt / GRIT_GEMM_DGELU       (also fc1's bias gradient, as slab sums)
            weight gradients, fc1 input gradient: library GEMMs (split-M, grit_amd/ops/linear.py)

Nothing else touches the [M, 4C] hidden map: no GELU, GeluBackward or column-sum kernel (4.8 + 2 ms of the 67 ms step)."""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend
from grit_amd.ops import gemm as G
from grit_amd.ops import layer_norm as LN
from grit_amd.ops import transposed as _transposed
from grit_amd.ops.linear import (WGRAD_STREAM, SlabGroup, column_sum, defer_weight_bias_grad, finish_group, fork, grad_slot, join,
                                 long_weight_grads_together, on_stream, single_use_now, slab_sum, weight_grad)
from grit_amd.ops.profiling import gemm_work, timed

MIN_ROWS = 2048


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _rows(t):
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.is_contiguous() else t2.contiguous()


def _mlp_backward(d_branch, n2, w1, pre, act, w2, needs, need_b2=False, group=None, params=None, row_scale=None, rows_per_sample=0):
    """Gradients of branch = fc2(gelu(fc1(n2))) w.r.t. (n2, w1, b1, w2[, b2]) given d_branch [M, C].  The chain of input
    gradients runs on the current stream, the weight / bias gradients beside it on the side stream (linear.fork).  `group`:
    the caller's SlabGroup -- every partial sum of the node (dW2, db1, dW1[, db2]) is then reduced by the caller's one launch
    (not combined with the side stream: the knob that enables it is an A/B diagnostic)."""
    need_x, need_w1, need_b1, need_w2 = needs
    if row_scale is not None and rows_per_sample > 0:
        backend.check_dropped_rows(d_branch, row_scale, "Mlp backward (drop-path factors)")
    d_n2 = d_w1 = d_b1 = d_w2 = d_b2 = None
    chain = need_x or need_w1 or need_b1
    side = fork(d_branch, act) if (chain and (need_w2 or need_b2)) else None
    if side is not None:
        group = None
    # params = (fc1.weight, fc2.weight) as Parameters when the module is declared single-use: inside a gradient-bucket scope the two
    # weight gradients then join the scope's grouped launch (ops.linear.defer_weight_bias_grad) instead of running here
    rs = (row_scale, rows_per_sample) if (row_scale is not None and rows_per_sample > 0) else None  # (the weight gradients skip dropped samples)
    dfr2 = defer_weight_bias_grad(d_branch, act, params[1], None, need_w2, False, True, row_scale=rs) if (params is not None and side is None) else None
    # both weight gradients of the node in one grouped launch (after d_pre exists): see linear.long_weight_grads_together
    together = dfr2 is None and side is None and group is not None and need_w2 and need_w1 and chain
    with on_stream(side):
        if dfr2 is not None:
            d_w2 = dfr2[0]
        elif need_w2 and not together:
            d_w2 = weight_grad(d_branch, act, group, param=w2, row_scale=rs)
        if need_b2:
            d_b2 = column_sum(d_branch, w2.dtype, group)
    if chain:
        w2t = _transposed.lookup(params[1] if params is not None else w2)  # made for all blocks at once by the backbone's forward
        # (row_scale: the drop-path factors already applied to the rows of d_branch -- tiles of dropped samples are zeros without a K loop)
        d_pre, partial = G.input_grad_dgelu(d_branch, w2t if w2t is not None else w2.t().contiguous(), pre, row_scale, rows_per_sample)
        join(side, d_w2, d_b2)
        side = fork(d_pre, partial, n2) if (need_x and (need_w1 or need_b1)) else None
        if side is not None:
            group = None
        with on_stream(side):
            if need_b1:
                d_b1 = (group.add(partial.unsqueeze(0), w1.dtype) if group is not None else slab_sum(partial.unsqueeze(0), w1.dtype))[0]
            dfr1 = defer_weight_bias_grad(d_pre, n2, params[0], None, need_w1, False, True, row_scale=rs) if (params is not None and side is None) else None
            parts = long_weight_grads_together([(d_branch, act), (d_pre, n2)], row_scale=rs) if (together and dfr1 is None) else None
            if parts is not None:
                outs = []
                for part, (wt, dyt, xt) in zip(parts, ((w2, d_branch, act), (w1, d_pre, n2))):
                    slot = grad_slot(wt, dyt.dtype, dyt.device)
                    outs.append(group.add(part.unsqueeze(0), dyt.dtype,
                                          out=None if slot is None else slot.view(1, dyt.shape[1], xt.shape[1]))[0])
                d_w2, d_w1 = outs
            else:
                if together:
                    d_w2 = weight_grad(d_branch, act, group, param=w2, row_scale=rs)
                if dfr1 is not None:
                    d_w1 = dfr1[0]
                elif need_w1:
                    d_w1 = weight_grad(d_pre, n2, group, param=w1, row_scale=rs)
        if need_x:
            d_n2 = G.long_input_grad(d_pre, params[0] if params is not None else w1)  # NT on fc1.weight^T (own kernel / library NT)
            if d_n2 is None:
                with timed("gemm_lib", **gemm_work(d_pre.shape[0], w1.shape[1], w1.shape[0])):
                    d_n2 = torch.mm(d_pre, w1)
        join(side, d_b1, d_w1)
    return d_n2, d_w1, d_b1, d_w2, d_b2


class _MlpFn(Function):
    """branch = fc2(gelu(fc1(x)))."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, single_use=False):
        ctx.params = (w1, b1, w2, b2) if single_use else None
        x2 = _rows(x)
        pre, act = G.linear_bias_gelu(x2, w1, b1)
        out = G.long_linear(act, w2, b2)  # (the stage-0 map: the own narrow-output kernel; None elsewhere)
        if out is None:
            with timed("gemm_lib", **gemm_work(act.numel() // act.shape[-1], w2.shape[0], w2.shape[1])):
                out = F.linear(act, w2, b2)
        ctx.save_for_backward(x2, w1, pre, act, w2)
        ctx.shape = x.shape
        return out.view(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, w1, pre, act, w2 = ctx.saved_tensors
        d_branch = _rows(dy)
        ni = ctx.needs_input_grad
        group = SlabGroup()
        ps = ctx.params
        d_x, d_w1, d_b1, d_w2, d_b2 = _mlp_backward(d_branch, x2, w1, pre, act, w2, (ni[0], ni[1], ni[2], ni[3]), ni[4], group,
                                                    params=None if ps is None else (ps[0], ps[2]))
        finish_group(group, ps is not None and not WGRAD_STREAM, [] if ps is None else
                     [(ps[0], d_w1), (ps[1], d_b1), (ps[2], d_w2), (ps[3], d_b2)])
        return (None if d_x is None else d_x.view(ctx.shape)), d_w1, d_b1, d_w2, d_b2, None


class _MlpAddLayerNormFn(Function):
    """(x_in, shortcut, scale) -> (x, LayerNorm(x)) with x = shortcut + scale[b] * fc2(gelu(fc1(x_in)))."""

    @staticmethod
    def forward(ctx, x_in, w1, b1, w2, b2, shortcut, scale, weight, bias, eps, single_use=False):
        ctx.params = (w1, b1, w2, b2, weight, bias) if single_use else None
        x2 = _rows(x_in)
        # (scale: the drop-path factors of this branch -- the fc1 tiles of samples it removes are not computed)
        pre, act = G.linear_bias_gelu(x2, w1, b1, scale, x2.shape[0] // shortcut.shape[0] if scale is not None else 0)
        C = shortcut.shape[-1]
        s2 = _rows(shortcut)
        rows = s2.shape[0]
        y = torch.empty_like(s2)
        mean = torch.empty(rows, dtype=torch.float32, device=s2.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=s2.device)
        xb, wb = int(s2.dtype == torch.bfloat16), int(weight.dtype == torch.bfloat16)
        # fc2 with the residual connection in its epilogue (x = shortcut + factor * branch, one launch), then the LayerNorm of x alone
        x = G.long_linear_residual(act, w2, b2, s2, scale, rows // shortcut.shape[0])
        if x is not None:
            with _lib.device_guard(s2.device):
                st = _lib.load().grit_layernorm_fwd(_ptr(x), _ptr(weight), _ptr(bias), rows, C, eps, xb, wb, _ptr(y), _ptr(mean), _ptr(rstd),
                                                    _lib.current_stream_ptr())
            _lib.check(st, "grit_layernorm_fwd")
        else:
            branch = G.long_linear(act, w2, b2)  # (the stage-0 map: the own narrow-output kernel; None elsewhere)
            if branch is None:
                with timed("gemm_lib", **gemm_work(act.numel() // act.shape[-1], w2.shape[0], w2.shape[1])):
                    branch = F.linear(act, w2, b2)
            x = torch.empty_like(s2)
            with _lib.device_guard(s2.device):
                st = _lib.load().grit_add_layernorm_fwd(_ptr(s2), _ptr(branch), _ptr(scale), rows // shortcut.shape[0], 0.0, None,
                                                        _ptr(weight), _ptr(bias), rows, C, eps, xb, wb, _ptr(x), _ptr(y),
                                                        _ptr(mean), _ptr(rstd), _lib.current_stream_ptr())
            _lib.check(st, "grit_add_layernorm_fwd")
        ctx.save_for_backward(x, weight, mean, rstd, scale, x2, w1, pre, act, w2)
        ctx.shape, ctx.in_shape = shortcut.shape, x_in.shape
        ctx.set_materialize_grads(False)
        return x.view(shortcut.shape), y.view(shortcut.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gy):
        x, weight, mean, rstd, scale, x2, w1, pre, act, w2 = ctx.saved_tensors
        group = SlabGroup()  # LayerNorm sums, dW2, db1, dW1: one reduction launch for the whole node
        dx, d_branch, sums = LN._add_layer_norm_backward(x, weight, mean, rstd, scale, gx, gy, ctx.shape[0], True, 0.0, None, group)
        ni = ctx.needs_input_grad
        ps = ctx.params
        d_x, d_w1, d_b1, d_w2, _ = _mlp_backward(d_branch, x2, w1, pre, act, w2, (ni[0], ni[1], ni[2], ni[3]), group=group,
                                                 params=None if ps is None else (ps[0], ps[2]), row_scale=scale,
                                                 rows_per_sample=x2.shape[0] // ctx.shape[0] if scale is not None else 0)
        # (sums in another dtype than the parameters would be converted -- read -- below, before a deferred launch has run)
        finish_group(group, ps is not None and not WGRAD_STREAM and sums.dtype == w2.dtype and ni[4] and ni[7] and ni[8],
                     [] if ps is None else [(ps[0], d_w1), (ps[1], d_b1), (ps[2], d_w2), (ps[3], sums[2]), (ps[4], sums[0]),
                                            (ps[5], sums[1])])
        return ((None if d_x is None else d_x.view(ctx.in_shape)), d_w1, d_b1, d_w2, sums[2].to(w2.dtype), dx.view(ctx.shape),
                None, sums[0], sums[1], None, None)


def _fits(x, mlp):
    fc1, fc2 = mlp.fc1, mlp.fc2
    return (backend.override() is None and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_autocast_enabled()
            and isinstance(mlp.act, torch.nn.GELU) and getattr(mlp.act, "approximate", "none") == "none"
            and (mlp.drop.p == 0. or not mlp.training)
            and fc1.bias is not None and fc2.bias is not None and fc1.weight.dtype == torch.bfloat16
            and fc2.weight.dtype == torch.bfloat16 and fc1.bias.dtype == torch.bfloat16
            and x.numel() // x.shape[-1] >= MIN_ROWS
            and fc1.weight.is_contiguous() and fc2.weight.is_contiguous()
            and G.supported(x.reshape(-1, x.shape[-1]), fc1.weight) and fc2.weight.shape[0] % 128 == 0
            and fc1.bias.data_ptr() % 16 == 0)


def mlp(x, module):
    """module(x) for a Swin Mlp; the fused node when it applies, the module itself otherwise."""
    if not _fits(x, module):
        return module(x)
    fc1, fc2 = module.fc1, module.fc2
    if not (torch.is_grad_enabled() and (x.requires_grad or fc1.weight.requires_grad or fc2.weight.requires_grad)):
        act = G.gemm_nt(_rows(x), fc1.weight, G.BIAS_GELU, bias=fc1.bias)  # frozen stage / inference: no pre-activation kept
        out = G.long_linear(act, fc2.weight, fc2.bias)  # (stage-0 map: the own narrow-output kernel)
        if out is None:
            out = F.linear(act, fc2.weight, fc2.bias)
        return out.view(x.shape[:-1] + (fc2.weight.shape[0],))
    return _MlpFn.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                        single_use_now(getattr(fc1, "single_use", False) and getattr(fc2, "single_use", False)))


def hidden(x, module):
    """module.hidden(x) = drop(gelu(fc1(x))); without autograd (inference, frozen stages) fc1 + bias + GELU is one launch."""
    fc1 = module.fc1
    if not _fits(x, module) or (torch.is_grad_enabled() and (x.requires_grad or fc1.weight.requires_grad)):
        return module.hidden(x)
    act = G.gemm_nt(_rows(x), fc1.weight, G.BIAS_GELU, bias=fc1.bias)
    return act.view(x.shape[:-1] + (fc1.weight.shape[0],))


def mlp_add_layer_norm(x_in, module, shortcut, scale, norm):
    """(x, norm(x)) with x = shortcut + scale[b] * module(x_in), or None when the fused node does not apply."""
    C = shortcut.shape[-1]
    ok = (_fits(x_in, module) and torch.is_grad_enabled()
          and (x_in.requires_grad or module.fc1.weight.requires_grad or module.fc2.weight.requires_grad)
          and isinstance(norm, LN.LayerNorm) and norm.elementwise_affine and len(norm.normalized_shape) == 1
          and C in LN.SUPPORTED_C and C <= 1024 and module.fc2.weight.shape[0] == C and shortcut.dtype == torch.bfloat16
          and norm.weight.dtype == norm.bias.dtype == torch.bfloat16 and x_in.shape[:-1] == shortcut.shape[:-1])
    if not ok:
        return None
    if scale is not None:
        scale = scale.reshape(-1).float().contiguous()
    fc1, fc2 = module.fc1, module.fc2
    return _MlpAddLayerNormFn.apply(x_in, fc1.weight, fc1.bias, fc2.weight, fc2.bias, shortcut, scale, norm.weight.contiguous(),
                                    norm.bias.contiguous(), float(norm.eps),
                                    single_use_now(getattr(fc1, "single_use", False) and getattr(fc2, "single_use", False)))
