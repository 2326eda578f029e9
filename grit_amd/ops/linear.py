"""nn.Linear with a backward arranged for MI355X on the long token maps of GRIT (M = 51 200 .. 272 000 rows).

The GEMMs stay library GEMMs (hipBLASLt through F.linear / torch.mm / torch.bmm -- plumbing); what changes is how the
weight gradient is posed.  dW = dY^T X reduces over M: one [N, K] output of at most a few hundred 256x256 tiles with a
51 200+-deep inner dimension leaves most of the 256 CUs idle or forces the library into slow split kernels (measured,
tuned: 203 us for 2048x512 @ M = 51 200, 452 us for 256x1024 @ M = 204 800).  Splitting M into S slabs and running ONE
batched GEMM with fp32 partials (bmm, out_dtype = float32) followed by a sum over the S partials fills the chip:
132 us and 136 us for the same problems (tools/bench_weight_grad.py).  The bias gradient db = colsum(dY) uses the streaming
column-sum kernel (grit_colsum).  Used for the four Linears of every Swin block and MSDeformAttn.value_proj."""
import ctypes
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from grit_amd import lib as _lib
from grit_amd.ops import backend
from grit_amd.ops.profiling import timed

MIN_ROWS = int(os.environ.get("GRIT_LINEAR_MIN_ROWS", "512"))  # below this the launch overhead dominates: leave it to torch

# The weight / bias gradients are not on the critical path of backward (nothing downstream of the node reads them), the input
# gradient is.  With GRIT_WGRAD_STREAM=1 they are enqueued on a second HIP stream and the input-gradient GEMM on the current
# one, so that the two GEMMs could fill each other's partial last wave of tiles and the small slab-sum / column-sum launches
# run beside a GEMM instead of between two.  MEASURED SLOWER (64.4 -> 69.3 ms/step, profiles/r02/negative_results.txt): two
# 128-KB-LDS GEMMs sharing the chip evict each other's L2 working set; the knob stays for A/B runs, default off.
WGRAD_STREAM = os.environ.get("GRIT_WGRAD_STREAM", "0") == "1"
_side_streams = {}


def fork(*inputs):
    """Side stream ordered after everything enqueued so far on the current stream (None when the knob is off / on CPU).
    `inputs` are the tensors the side work reads: their memory is not handed out again before that work has run."""
    if not (WGRAD_STREAM and inputs and inputs[0].is_cuda):
        return None
    dev = inputs[0].device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    for t in inputs:
        t.record_stream(side)
    return side


def join(side, *outputs):
    """The current stream waits for the side work; `outputs` (allocated on the side stream) are consumed on the current one."""
    if side is None:
        return
    main = torch.cuda.current_stream(side.device)
    main.wait_stream(side)
    for t in outputs:
        if t is not None:
            t.record_stream(main)


class on_stream:
    """`with on_stream(side):` -- torch.cuda.stream(side), or nothing when side is None."""

    def __init__(self, side):
        self.ctx = torch.cuda.stream(side) if side is not None else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def slab_sum(partial, out_dtype, slabs=None):
    """f32 partial sums [groups, slabs_allocated, n...] (contiguous) -> [groups, n...] in out_dtype: the sum over the first
    `slabs` slabs of every group and the dtype cast in one launch (grit_slab_sum)."""
    slabs = partial.shape[1] if slabs is None else slabs
    tail = partial.shape[2:]
    n = 1
    for d in tail:
        n *= d
    out = torch.empty((partial.shape[0],) + tuple(tail), dtype=out_dtype, device=partial.device)
    with _lib.device_guard(partial.device):
        st = _lib.load().grit_slab_sum(ctypes.c_void_p(partial.data_ptr()), partial.shape[0], partial.stride(0), slabs, n,
                                       ctypes.c_void_p(out.data_ptr()), int(out_dtype == torch.bfloat16), _lib.current_stream_ptr())
    _lib.check(st, "grit_slab_sum")
    return out


def column_sum(x2d, out_dtype=torch.float32):
    """[M, N] (bf16 / f32, contiguous, N % 8 == 0) -> [N] in out_dtype (f32 accumulation)."""
    M, N = x2d.shape
    strips = max(1, (N + 511) // 512)
    slabs = max(1, min(256, (M * N) // (1 << 18), 2048 // strips))
    # small maps (the decoders' M = 640 .. 4 800 rows): enough slabs to put a workgroup on every CU, >= 16 rows each
    slabs = max(slabs, min(-(-256 // strips), M // 16, 256))
    partial = torch.empty(slabs, N, dtype=torch.float32, device=x2d.device)
    with _lib.device_guard(x2d.device):
        st = _lib.load().grit_colsum(ctypes.c_void_p(x2d.data_ptr()), M, N, int(x2d.dtype == torch.bfloat16), slabs,
                                     ctypes.c_void_p(partial.data_ptr()), _lib.current_stream_ptr())
    _lib.check(st, "grit_colsum")
    return slab_sum(partial.unsqueeze(0), out_dtype)[0]


_SLAB_ROWS = int(os.environ.get("GRIT_WGRAD_SLAB_ROWS", "3200"))  # tuning knob (tools/bench_weight_grad.py)


def split_k(M):
    """Number of row slabs for the weight-gradient GEMM: ~3 200-6 400 rows per slab, at most 64, dividing M; powers of two
    where possible (other counts run 20-50 % slower in the library).  Swin stage 3 (M = 12 800): 4 slabs
    (tools/bench_weight_grad_slabs.py: 1024x1024 80 -> 43 us, 3072x1024 116 -> 87 us)."""
    if M < 25600:
        return 4 if (M >= 6400 and M % 4 == 0) else 1
    s = min(64, M // _SLAB_ROWS)
    while s > 1 and M % s:
        s -= 1
    return s


def weight_grad(dy2, x2):
    """dW [N, K] = dy2^T [N, M] @ x2 [M, K], split over M into one batched GEMM with fp32 partial sums."""
    M, N = dy2.shape
    S = split_k(M) if (dy2.is_cuda and dy2.dtype == torch.bfloat16) else 1
    with timed("gemm_lib", flops=2.0 * M * N * x2.shape[1]):
        if S == 1:
            return torch.mm(dy2.t(), x2)
        part = torch.bmm(dy2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, x2.shape[1]), out_dtype=torch.float32)
    return slab_sum(part.unsqueeze(0), dy2.dtype)[0]


class _LinearFn(Function):

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        with timed("gemm_lib", flops=2.0 * x.numel() * weight.shape[0]):
            return F.linear(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        dx = dw = db = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        side = fork(dy2, x2) if (ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or need_b)) else None
        with on_stream(side):
            if ctx.needs_input_grad[1]:
                dw = weight_grad(dy2, x2)
            if need_b:
                db = column_sum(dy2, weight.dtype)
        if ctx.needs_input_grad[0]:
            with timed("gemm_lib", flops=2.0 * dy2.numel() * weight.shape[1]):
                dx = torch.mm(dy2, weight).view(x.shape)
        join(side, dw, db)
        return dx, dw, db


class _SharedInputLinearsFn(Function):
    """y_l = x W_l^T + b_l for n Linear layers that read the SAME input (the six value_proj of the deformable decoder all
    read the flat feature map).  Left to autograd, dx is built as n GEMMs plus n - 1 full-size additions (5 x 3 x 278 MB of
    traffic at batch 32); here the later GEMMs accumulate into the first one's output (addmm, beta = 1)."""

    @staticmethod
    def forward(ctx, x, n, *params):
        weights, biases = params[:n], params[n:]
        ctx.save_for_backward(x, *weights)
        ctx.n = n
        ctx.set_materialize_grads(False)  # an unused output arrives as None, not as a zero map to multiply
        with timed("gemm_lib", flops=2.0 * x.numel() * weights[0].shape[0] * n):
            return tuple(F.linear(x, w, b) for w, b in zip(weights, biases))

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        x, weights = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        n = ctx.n
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        dx2, dws, dbs = None, [None] * n, [None] * n
        for l in range(n):
            if dys[l] is None:
                continue
            dy2 = dys[l].reshape(-1, dys[l].shape[-1])
            dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
            if ctx.needs_input_grad[0]:
                with timed("gemm_lib", flops=2.0 * dy2.numel() * weights[l].shape[1]):
                    if dx2 is None:
                        dx2 = torch.mm(dy2, weights[l])
                    else:
                        dx2.addmm_(dy2, weights[l])
            if ctx.needs_input_grad[2 + l]:
                dws[l] = weight_grad(dy2, x2)
            if ctx.needs_input_grad[2 + n + l]:
                dbs[l] = column_sum(dy2, weights[l].dtype)
        dx = None if dx2 is None else dx2.view(x.shape)
        return (dx, None) + tuple(dws) + tuple(dbs)


def shared_input_linears(x, linears):
    """[lin(x) for lin in linears] with one fused input gradient; falls back to the modules themselves when the fused
    node does not apply (CPU / oracle runs, no grad, mixed dtypes, missing bias)."""
    fits = (backend.override() is None and x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and len(linears) > 1 and all(lin.bias is not None and lin.weight.dtype == x.dtype
                                         and lin.weight.shape == linears[0].weight.shape for lin in linears)
            and x.dtype in (torch.bfloat16, torch.float32) and linears[0].weight.shape[0] % 8 == 0
            and x.numel() // x.shape[-1] >= MIN_ROWS)
    if not fits:
        return [lin(x) for lin in linears]
    return list(_SharedInputLinearsFn.apply(x, len(linears), *[lin.weight for lin in linears], *[lin.bias for lin in linears]))


def linear(x, weight, bias):
    fits = (backend.override() is None and x.is_cuda and torch.is_grad_enabled()
            and not torch.is_autocast_enabled()
            and (x.requires_grad or weight.requires_grad) and x.dtype == weight.dtype
            and x.dtype in (torch.bfloat16, torch.float32) and weight.shape[0] % 8 == 0
            and x.numel() // x.shape[-1] >= MIN_ROWS)
    if not fits:
        return F.linear(x, weight, bias)
    return _LinearFn.apply(x, weight, bias)


class Linear(nn.Linear):
    """Same parameters / state-dict keys as nn.Linear."""

    def forward(self, input):
        return linear(input, self.weight, self.bias)
