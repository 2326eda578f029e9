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
from grit_amd.ops.profiling import gemm_work, timed

MIN_ROWS = int(os.environ.get("GRIT_LINEAR_MIN_ROWS", "512"))  # below this the launch overhead dominates: leave it to torch

# The weight / bias gradients are not on the critical path of backward (nothing downstream of the node reads them), the input
# gradient is.  With GRIT_WGRAD_STREAM=1 they are enqueued on a second HIP stream and the input-gradient GEMM on the current
# one, so that the two GEMMs could fill each other's partial last wave of tiles and the small slab-sum / column-sum launches
# run beside a GEMM instead of between two.  MEASURED SLOWER (64.4 -> 69.3 ms/step, profiles/r02/negative_results.txt): two
# 128-KB-LDS GEMMs sharing the chip evict each other's L2 working set; the knob stays for A/B runs, default off.
# (round 6, same-box A/B inside the captured step, profiles/r06/ab_wgrad_stream.txt: 47.5 -> 48.5 ms; with the join deferred to the bucket
# pack -- the weight gradients beside the HBM-bound LayerNorm / window-attention kernels that follow -- 48.2 ms: matrix work and HBM
# streaming do not overlap on this chip, they add up to ~80 % of their sum, tools/micro/overlap_probe.py)
WGRAD_STREAM = os.environ.get("GRIT_WGRAD_STREAM", "0") == "1"
_side_streams = {}


# Small maps (the two decoders and the grid net: 640 .. 4 800 rows) are another matter: their GEMMs fill a few dozen CUs for
# 10-30 us each and the backward pass is one long dependent chain of them, so the weight / bias gradients -- nothing downstream
# in backward reads them -- can run BESIDE the chain on a second stream.  MEASURED SLOWER in the benchmark step (65.0 -> 65.7 ms,
# two alternating passes on one box, profiles/r03/negative_results.txt): that part of the step is bound by how fast the host
# enqueues its tiny kernels, and every fork adds two cross-stream event operations.  GRIT_WGRAD_STREAM_SMALL=1 (default 0) enables
# it for nodes with fewer than SMALL_ROWS rows, and only while a gradient-bucket wrapper has declared a deferral scope
# (grit_amd.ddp: begin_deferral in forward, wait_deferred before it packs or hands out gradients): the main stream then does not
# wait for the side stream at the end of the node but when the gradients are first consumed.  A parameter used twice in one
# forward pass would have its two gradients added by autograd on the main stream before that point, so the deferral is opt-in
# per call site (`single_use=True`: the caller guarantees the weight receives exactly one gradient per backward pass).
WGRAD_STREAM_SMALL = os.environ.get("GRIT_WGRAD_STREAM_SMALL", "0") == "1"
SMALL_ROWS = int(os.environ.get("GRIT_WGRAD_SMALL_ROWS", "16384"))
_deferral = {"active": False, "pending": set()}


def mark_single_use(*modules):
    """Declare that every Linear inside `modules` is applied exactly once per forward pass (its weight receives exactly one
    gradient per backward pass).  Model constructors call this for the layers of the two decoders and the grid net."""
    for mod in modules:
        for m in mod.modules():
            if isinstance(m, nn.Linear):
                m.single_use = True


class suspend_single_use(object):
    """`with suspend_single_use():` -- the declarations above do not hold inside (step-wise decoding WITH gradient applies the
    caption decoder once per generated token: self-critical training)."""

    def __enter__(self):
        self.prev = _deferral.get("suspended", False)
        _deferral["suspended"] = True

    def __exit__(self, *exc):
        _deferral["suspended"] = self.prev


def single_use_now(flag):
    """The declaration as it holds for the forward call being recorded (nodes store this in their context: the backward pass
    runs outside any suspend_single_use block)."""
    return bool(flag) and not _deferral.get("suspended", False)


# GRIT_WGRAD_DEFER (default 1): inside a deferral scope the weight / bias gradients of single-use Linears on short maps are not
# computed by their node at all.  The node returns EMPTY gradient tensors, remembers (dY, X, where the results go), and the
# scope's owner -- the gradient-bucket wrapper, the only consumer of parameter gradients between backward and the optimizer --
# has them all computed by ONE grouped launch (grit_wgrad_small_grouped + one grouped slab sum) right before it first reads them
# (ddp._pack / finish_gradient_sync -> flush_deferred).  ~95 library GEMMs of 24-40 us, ~60 column-sum and ~90 reduction launches
# per step become a handful of launches that fill the chip.  Safety: a job is only deferred when the parameter has no gradient
# yet; at flush time the parameter's .grad must still be the very tensor the node returned (autograd keeps -- "steals" -- a fresh
# contiguous gradient), otherwise the parameter received a second gradient and the run stops with an error instead of training
# on garbage.
WGRAD_DEFER = os.environ.get("GRIT_WGRAD_DEFER", "1") != "0"
# GRIT_WGRAD_DEFER_LONG=1 (default 0): the long maps' weight gradients too (where single-use).  Measured +0.45 ms
# (profiles/r03/negative_results.txt #14): fewer and fatter slices, but operands that were in the Infinity Cache when their node ran
# come back from HBM a bucket later.
WGRAD_DEFER_LONG = os.environ.get("GRIT_WGRAD_DEFER_LONG", "0") == "1"
_deferral["jobs"] = []
_deferral["slabs"] = []
# GRIT_WGRAD_PARK (default 1): a long-map weight gradient whose Linear is marked with park_weight_grad_for_partner (the attention
# output projection of a Swin block) is not launched by its own node: it waits -- two kernels -- for the weight gradient of the
# partner that backward reaches next (the qkv Linear of the same block) and runs in THAT node's launch of the long-map kernel.
# Alone the projection is 4 output tiles cut into 64 row slices (64 MB of fp32 partials for a 0.5 MB gradient) and qkv 12 tiles in
# 21 slices; together they are 16 tiles in 16 slices of 100 steps: half the partial bytes to write and to sum, one launch less.
WGRAD_PARK = os.environ.get("GRIT_WGRAD_PARK", "1") != "0"
_deferral["parked"] = []


def park_weight_grad_for_partner(first, partner):
    """first, partner: Linear modules; backward reaches `first` (its weight gradient is parked) shortly before `partner` (whose node
    launches both).  Both must run once per forward pass; `first` must be single-use (mark_single_use)."""
    first.weight._grit_wgrad_park = True
    partner.weight._grit_wgrad_pickup = True


def defer_weight_bias_grad(dy2, x2, weight, bias, need_dw, need_db, single_use, row_scale=None):
    """(dW, db) as empty tensors that flush_deferred() will fill, or None when the job must be done by the node itself."""
    if not (WGRAD_DEFER and single_use and need_dw and _deferral["active"] and dy2.is_cuda and dy2.dtype == torch.bfloat16
            and x2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.grad is None
            and (not need_db or (bias is not None and bias.grad is None and bias.dtype == torch.bfloat16))
            and not backend.foreign_capture()):
        return None
    M, N = dy2.shape
    K = x2.shape[1]
    # short maps always; long maps (the Swin blocks) when the long-map kernel takes the shape: the ten or so weight gradients a
    # gradient bucket collects then run as ONE launch of ~256 workgroups with 800-step loops and TWO row slices each instead of ten
    # launches with 16-64 slices -- the fp32 slices (3.9 GB per step written and read back) shrink by an order of magnitude
    long_ok = (WGRAD_DEFER_LONG and WGRAD_TN_GROUPED and M >= WGRAD_SMALL_MAX_ROWS
               and _lib.load().grit_wgrad_tn_group_ok(M, N, K) == 1)
    park = (WGRAD_PARK and WGRAD_TN and WGRAD_TN_GROUPED and not long_ok and not need_db and M >= WGRAD_SMALL_MAX_ROWS
            and getattr(weight, "_grit_wgrad_park", False) and _lib.load().grit_wgrad_tn_group_ok(M, N, K) == 1)
    long_ok = long_ok or park
    if not ((M < WGRAD_SMALL_MAX_ROWS or long_ok) and N % 64 == 0 and K % 64 == 0 and dy2.stride(1) == 1 and x2.stride(1) == 1
            and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0):
        return None
    # straight into the gradient bucket (round 5: the short maps' gradients too -- ~60 M decoder parameters were copied into their slots
    # by the bucket pack, 12 multi-tensor launches of ~19 us per step; GRIT_DEFER_SLOT_SHORT=0 restores the copies)
    dw = grad_slot(weight, torch.bfloat16, dy2.device) if (long_ok or DEFER_SLOT_SHORT) else None
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.bfloat16, device=dy2.device)
    db = (grad_slot(bias, torch.bfloat16, dy2.device) if DEFER_SLOT_SHORT else None) if need_db else None
    if need_db and db is None:
        db = torch.empty((N,), dtype=torch.bfloat16, device=dy2.device)
    # no reference to dw / db is kept (autograd only adopts a gradient tensor nobody else holds): addresses only
    _deferral["parked" if park else "jobs"].append((dy2, x2, weight, bias if need_db else None, dw.data_ptr(),
                                                    db.data_ptr() if need_db else 0, M, N, K, row_scale))
    return dw, db


def defer_packed_weight_bias_grad(parts, weight, bias):
    """A packed projection (nn.MultiheadAttention.in_proj_weight / in_proj_bias: the rows of several Linears in one parameter) whose
    row ranges are computed from DIFFERENT inputs: parts = [(dy2 [M, N_i], x2 [M, K]), ...] in row order.  Returns (dW [sum N_i, K],
    db [sum N_i]) as tensors that flush_deferred() fills range by range -- each part is one more problem of the scope's grouped launch --
    or None when the jobs must be done by the node itself (same conditions as defer_weight_bias_grad; all or nothing)."""
    if not (WGRAD_DEFER and _deferral["active"] and weight.grad is None and bias is not None and bias.grad is None
            and weight.dtype == torch.bfloat16 and bias.dtype == torch.bfloat16 and not backend.foreign_capture()):
        return None
    K = weight.shape[1]
    for dy2, x2 in parts:
        M, N = dy2.shape
        if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and M < WGRAD_SMALL_MAX_ROWS and N % 64 == 0
                and K % 64 == 0 and x2.shape == (M, K) and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0
                and x2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0):
            return None
    if sum(dy2.shape[1] for dy2, _ in parts) != weight.shape[0]:
        return None
    dev = parts[0][0].device
    dw = grad_slot(weight, torch.bfloat16, dev) if DEFER_SLOT_SHORT else None
    if dw is None:
        dw = torch.empty(tuple(weight.shape), dtype=torch.bfloat16, device=dev)
    db = grad_slot(bias, torch.bfloat16, dev) if DEFER_SLOT_SHORT else None
    if db is None:
        db = torch.empty(tuple(bias.shape), dtype=torch.bfloat16, device=dev)
    pw0, pb0, row = dw.data_ptr(), db.data_ptr(), 0
    for dy2, x2 in parts:
        M, N = dy2.shape
        _deferral["jobs"].append((dy2, x2, (weight, pw0), (bias, pb0), pw0 + row * K * 2, pb0 + row * 2, M, N, K, None))
        row += N
    return dw, db


def _short_transposed(weight, dy2):
    """W^T of the packed in-projection [3E, E] from grit_amd.ops.transposed when both of its column ranges fit the short-map policy of
    grit_amd/ops/gemm.py (input gradients as NT products on the 64 x 64 x 64 tiles), else None."""
    if weight is None or not (dy2.is_cuda and dy2.dtype == torch.bfloat16):
        return None
    from grit_amd.ops import gemm as _gemm
    from grit_amd.ops import transposed
    E = weight.shape[1]
    if not (_gemm.OWN and _gemm.prefers_own_short(dy2.shape[0], E, 2 * E) and _gemm.prefers_own_short(dy2.shape[0], E, E)):
        return None
    wt = transposed.lookup(weight)
    return wt if (wt is not None and wt.shape == (E, 3 * E)) else None


class _PackedInProjFn(Function):
    """(qk_in, v_in, W [3E, E], b [3E]) -> (qk_in W[:2E]^T + b[:2E], v_in W[2E:]^T + b[2E:]): the in-projections of an
    nn.MultiheadAttention whose query / key input differs from its value input (DeformableTransformerDecoderLayer: q = k = tgt + pos,
    v = tgt; reference models/detection/det_module.py:313-326).  Taken apart with split() the weight gradient came back as two library
    TN GEMMs (41 + 31 us at 4 800 rows), two column-sum launches and three concatenations per layer; as ONE node over the packed
    parameter both row ranges join the gradient bucket's grouped weight-gradient launch and land in the parameter's bucket slot."""

    @staticmethod
    def forward(ctx, qk_in, v_in, weight, bias):
        E = weight.shape[1]
        ctx.save_for_backward(qk_in, v_in, weight)
        ctx.bias_param, ctx.weight_param = bias, weight
        qk = _own_linear(qk_in, weight[:2 * E], bias[:2 * E])  # (short maps: the own 64 x 64 x 64 tiles, grit_amd/ops/gemm.py)
        v = _own_linear(v_in, weight[2 * E:], bias[2 * E:]) if qk is not None else None
        if v is not None:
            return qk, v
        with timed("gemm_lib", **gemm_work(qk_in.numel() // E + v_in.numel() // E, 3 * E // 2, E)):
            return F.linear(qk_in, weight[:2 * E], bias[:2 * E]), F.linear(v_in, weight[2 * E:], bias[2 * E:])

    @staticmethod
    @once_differentiable
    def backward(ctx, dqk, dv):
        qk_in, v_in, weight = ctx.saved_tensors
        E = weight.shape[1]
        dqk2, dv2 = dqk.reshape(-1, 2 * E), dv.reshape(-1, E)
        dqk2 = dqk2 if dqk2.is_contiguous() else dqk2.contiguous()
        dv2 = dv2 if dv2.is_contiguous() else dv2.contiguous()
        q2, v2 = qk_in.reshape(-1, E), v_in.reshape(-1, E)
        q2 = q2 if q2.is_contiguous() else q2.contiguous()
        v2 = v2 if v2.is_contiguous() else v2.contiguous()
        d_qk_in = d_v_in = dw = db = None
        wt = _short_transposed(ctx.weight_param, dqk2)  # [E, 3E] = W^T, or None: no copy / outside the short-map policy
        if wt is not None:
            from grit_amd.ops import gemm as _gemm
            if ctx.needs_input_grad[0]:
                d_qk_in = _gemm.gemm_nt(dqk2, wt[:, :2 * E], _gemm.NONE, variant=_gemm.SHORT).view(qk_in.shape)
            if ctx.needs_input_grad[1]:
                d_v_in = _gemm.gemm_nt(dv2, wt[:, 2 * E:], _gemm.NONE, variant=_gemm.SHORT).view(v_in.shape)
        else:
            with timed("gemm_lib", **gemm_work(dqk2.shape[0] + dv2.shape[0], E, 3 * E // 2)):
                if ctx.needs_input_grad[0]:
                    d_qk_in = torch.mm(dqk2, weight[:2 * E]).view(qk_in.shape)
                if ctx.needs_input_grad[1]:
                    d_v_in = torch.mm(dv2, weight[2 * E:]).view(v_in.shape)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            deferred = defer_packed_weight_bias_grad([(dqk2, q2), (dv2, v2)], ctx.weight_param, ctx.bias_param) \
                if (ctx.needs_input_grad[2] and ctx.needs_input_grad[3]) else None
            if deferred is not None:
                dw, db = deferred
            else:
                group = SlabGroup() if dqk2.is_cuda else None
                dws = [weight_grad(dqk2, q2, group), weight_grad(dv2, v2, group)]
                dbs = [column_sum(dqk2, weight.dtype, group), column_sum(dv2, weight.dtype, group)] if dqk2.is_cuda \
                    else [dqk2.sum(0), dv2.sum(0)]
                if group is not None:
                    group.run()  # (the group's outputs hold their values only now)
                dw, db = torch.cat(dws), torch.cat(dbs)
        return d_qk_in, d_v_in, dw, db


def packed_in_proj(qk_in, v_in, weight, bias):
    """(qk [.., 2E], v [.., E]) of an nn.MultiheadAttention's packed in-projection with q = k = qk_in and v = v_in; the fused node on the
    device in training, the plain slices otherwise."""
    E = weight.shape[1]
    fits = (backend.override() is None and qk_in.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and weight.requires_grad and bias is not None and qk_in.dtype == v_in.dtype == weight.dtype == bias.dtype
            and qk_in.dtype in (torch.bfloat16, torch.float32) and weight.shape[0] == 3 * E and qk_in.shape == v_in.shape
            and qk_in.numel() // E >= MIN_ROWS)
    if not fits:
        return F.linear(qk_in, weight[:2 * E], bias[:2 * E]), F.linear(v_in, weight[2 * E:], bias[2 * E:])
    return _PackedInProjFn.apply(qk_in, v_in, weight, bias)


def defer_slab_group(group, checks):
    """Leave the pending reductions of `group` (a SlabGroup whose outputs are parameter gradients of a single-use node: LayerNorm
    dgamma / dbeta, a projection's bias gradient) to flush_deferred() instead of launching them now.  checks: [(parameter, device
    address its gradient must have at flush time)].  False (nothing deferred) when the deferral does not apply."""
    if not (WGRAD_DEFER and _deferral["active"] and group.jobs and not backend.foreign_capture()
            and all(p is not None and p.grad is None for p, _ in checks)):
        return False
    _deferral["slabs"].append((group.jobs, group.keep, checks))
    group.jobs, group.keep = [], []
    return True


_SECOND_GRADIENT = ("deferred %s: a parameter of a node declared single-use %s hold the gradient tensor the node returned (it "
                    "received a second gradient in the same backward pass, or none at all); do not mark the module single_use "
                    "(grit_amd.ops.linear.mark_single_use) or run with GRIT_WGRAD_DEFER=0")
_deferral["unverified"] = []


def _verify(p, ptr, what, final):
    """The parameter's .grad must be the very tensor the node returned.  A flush is triggered from a parameter's post-accumulate
    hook, i.e. possibly BETWEEN the deliveries of two gradients of one node: a gradient the engine has not delivered yet (it
    holds the tensor, the memory is valid) is written anyway and checked at the next flush -- before anything can have replaced
    it, because every bucket pack starts with a flush -- and at the final flush (backward is over) it must be there."""
    if isinstance(p, tuple):  # (parameter, address its gradient must have): a job that fills a ROW RANGE of a packed parameter's gradient
        p, ptr = p
    if p.grad is None and not final:
        _deferral["unverified"].append((p, ptr, what))
    elif p.grad is None or p.grad.data_ptr() != ptr:
        raise _lib.GritHipError(_SECOND_GRADIENT % (what, "does not" if p.grad is None else "no longer"))


def _verify_earlier(final):
    earlier, _deferral["unverified"] = _deferral["unverified"], []
    for p, ptr, what in earlier:
        _verify(p, ptr, what, final)


# GRIT_SLAB_DEFER_LONG=1 (default 0): also the LONG-map nodes (Swin blocks) leave their reductions to the scope's flush.  Measured
# neutral (59.26 / 59.23 -> 59.43 / 59.10 ms, profiles/r03/negative_results.txt): summed right behind the GEMM the fp32 partials
# are still in the Infinity Cache, summed a bucket later they come from HBM -- what the bigger launch saves, the colder data costs.
FINISH_DEFER = os.environ.get("GRIT_SLAB_DEFER_LONG", "0") == "1"


def finish_group(group, single_use, outputs):
    """End of a backward node that owns a SlabGroup.  outputs: [(parameter, gradient tensor the group's launch will fill)].  Inside a
    deferral scope, when every one of them belongs to a parameter declared single-use that holds no gradient yet, the launch is
    left to the scope's flush (one grouped launch for the nodes of a whole gradient bucket instead of one small launch per node:
    the long-map weight-gradient partials are then summed at streaming bandwidth); otherwise it runs now."""
    if group is None or not group.jobs:
        return
    checks = [(p, t.data_ptr()) for p, t in outputs if t is not None]
    if FINISH_DEFER and single_use and checks and all(p is not None for p, _ in checks) and defer_slab_group(group, checks):
        return
    group.run()


def _flush_deferred_slabs(into=None, final=False):
    slabs = _deferral["slabs"]
    if not slabs:
        return
    _deferral["slabs"] = []
    group = into if into is not None else SlabGroup()
    for jobs, keep, checks in slabs:
        for p, ptr in checks:
            _verify(p, ptr, "reduction", final)
        group.jobs.extend(jobs)
        group.keep.extend(keep)
        group.device = keep[0].device
    if into is None:
        group.run()


def flush_deferred(final=False):
    """Compute every deferred weight / bias gradient (one grouped GEMM launch + one grouped reduction per <= 32 problems) and
    every deferred reduction.  final: backward is over (finish_gradient_sync), every gradient must have been delivered."""
    _verify_earlier(final)
    jobs = _deferral["jobs"] + _deferral["parked"]  # parked jobs whose partner never came (a bucket boundary, a frozen partner)
    if not jobs:
        _flush_deferred_slabs(final=final)
        return
    _deferral["jobs"], _deferral["parked"] = [], []
    lib = _lib.load()
    dev = jobs[0][0].device
    for dy2, x2, w, b, pw, pb, M, N, K, _rs in jobs:
        _verify(w, pw, "weight gradient [%d, %d]" % (N, K), final)
        if b is not None:
            _verify(b, pb, "bias gradient [%d]" % N, final)
    # problems the long-map kernel takes (256 x 256 tiles) go through it TOGETHER: one of them alone is four tiles, thirty of them
    # are a launch of 256 workgroups with 75-step loops (1 PFLOP/s instead of the L2-bound 0.26 of the 64 x 64-tile kernel)
    big = [j for j in jobs if WGRAD_TN_GROUPED and lib.grit_wgrad_tn_group_ok(j[6], j[7], j[8])]
    small = [j for j in jobs if not (WGRAD_TN_GROUPED and lib.grit_wgrad_tn_group_ok(j[6], j[7], j[8]))]
    longs = [j for j in big if j[6] >= WGRAD_SMALL_MAX_ROWS]  # their own launches: a workgroup of theirs runs ~10x longer
    big = [j for j in big if j[6] < WGRAD_SMALL_MAX_ROWS]
    chunks = [("tn", longs[i:i + _lib.WGRAD_GROUP_MAX]) for i in range(0, len(longs), _lib.WGRAD_GROUP_MAX)]
    chunks += [("tn", big[i:i + _lib.WGRAD_GROUP_MAX]) for i in range(0, len(big), _lib.WGRAD_GROUP_MAX)]
    chunks += [("small", small[i:i + _lib.WGRAD_GROUP_MAX]) for i in range(0, len(small), _lib.WGRAD_GROUP_MAX)]
    with _lib.device_guard(dev):
        for ci, (kind, chunk) in enumerate(chunks):
            if kind == "tn":
                tiles = sum((j[7] // 256) * (j[8] // 256) for j in chunk)
                best, best_fill = 1, 0.0
                for cand in range(1, 9):  # row slices per problem: the fullest last round of workgroups, loops of >= 16 steps
                    wgs = sum((j[7] // 256) * (j[8] // 256) * max(1, min(cand, (j[6] // 32) // 16)) for j in chunk)
                    fill = wgs / (-(-wgs // 256) * 256.0)
                    if fill > best_fill + 0.02:
                        best, best_fill = cand, fill
                splits = [tn_slices(j[6], max(1, min(best, (j[6] // 32) // 16))) for j in chunk]
                slabs = splits if WGRAD_TN_BIAS else [max(1, min(256, j[6] // 64)) for j in chunk]  # bias: by-product per slice
            else:
                splits = [lib.grit_wgrad_group_splits(j[6]) for j in chunk]
                slabs = splits
            sizes, total = [], 0
            for job, S, sl in zip(chunk, splits, slabs):
                dy2, x2, w, b, pw, pb, M, N, K, _rs = job
                sizes.append((S, sl, total, total + S * N * K))
                total += S * N * K + (sl * N if b is not None else 0)
            work = torch.empty(total, dtype=torch.float32, device=dev)
            base = work.data_ptr()
            table = (_lib.WgradJob * len(chunk))()
            ctable = (_lib.ColsumJob * len(chunk))()
            nc = 0
            group = SlabGroup()
            for t, (job, (S, sl, woff, boff)) in enumerate(zip(chunk, sizes)):
                dy2, x2, w, b, pw, pb, M, N, K, rs = job
                bias_here = (base + 4 * boff) if b is not None else None
                table[t] = _lib.WgradJob(dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), M, N, K, S, base + 4 * woff,
                                         None if (kind == "tn" and not WGRAD_TN_BIAS) else bias_here, *_rows_arg(rs, M))
                if kind == "tn" and b is not None and not WGRAD_TN_BIAS:
                    ctable[nc] = _lib.ColsumJob(dy2.data_ptr(), dy2.stride(0), M, N, sl, bias_here)
                    nc += 1
                group.add_raw(work[woff:], 1, 0, S, N * K, pw, True)
                if b is not None:
                    group.add_raw(work[boff:], 1, 0, sl, N, pb, True)
            with timed("gemm_own", flops=2.0 * sum(j[6] * j[7] * j[8] for j in chunk), kernel="wgrad_tn" if kind == "tn" else "wgrad_small",
                       bytes=sum(2.0 * j[6] * (j[7] + j[8]) + 4.0 * S * j[7] * j[8] for j, S in zip(chunk, splits))):
                if kind == "tn":
                    st = lib.grit_wgrad_tn_grouped(table, len(chunk), _lib.current_stream_ptr())
                else:
                    st = lib.grit_wgrad_small_grouped(table, len(chunk), _lib.current_stream_ptr())
            _lib.check(st, "grit_wgrad_tn_grouped" if kind == "tn" else "grit_wgrad_small_grouped")
            if nc:
                _lib.check(lib.grit_colsum_grouped(ctable, nc, _lib.current_stream_ptr()), "grit_colsum_grouped")
            if ci == len(chunks) - 1:
                _flush_deferred_slabs(into=group, final=final)  # the nodes' own deferred reductions ride in the last chunk's launch
            group.run()  # keeps `work` (and through `chunk` the operands) alive until the launches are enqueued


def begin_deferral(owner=None):
    """owner: the gradient-bucket wrapper that will call end_deferral() (finish_gradient_sync).  A scope left open by ANOTHER
    wrapper -- its pass ended with an exception, or its backward never ran -- is abandoned, not inherited."""
    if not _deferral["active"] or _deferral.get("owner") is not owner:
        abandon_deferred()
    _deferral["active"] = True
    _deferral["owner"] = owner


def close_deferral(owner=None):
    """A forward pass of `owner` that records no graph: whatever scope it had open is over (nothing will flush it)."""
    if _deferral["active"] and _deferral.get("owner") is owner:
        abandon_deferred()
        _deferral["active"] = False


def abandon_deferred():
    """Forget every pending job (after an exception: the gradients of that pass are void anyway)."""
    _deferral["jobs"], _deferral["slabs"], _deferral["unverified"], _deferral["parked"] = [], [], [], []


def wait_deferred(final=False):
    """Deferred weight gradients are computed now, and the current stream waits for every side stream that still runs
    weight-gradient work: call before anything reads parameter gradients."""
    flush_deferred(final)
    for side in _deferral["pending"]:
        torch.cuda.current_stream(side.device).wait_stream(side)
    _deferral["pending"].clear()


def end_deferral():
    try:
        wait_deferred(final=True)
    finally:
        _deferral["active"] = False


def fork(*inputs, rows=None, single_use=False):
    """Side stream ordered after everything enqueued so far on the current stream (None when the knob is off / on CPU).
    `inputs` are the tensors the side work reads: their memory is not handed out again before that work has run.
    rows / single_use: the small-map deferral described above."""
    if not (inputs and inputs[0].is_cuda):
        return None
    small = (WGRAD_STREAM_SMALL and single_use and rows is not None and rows < SMALL_ROWS and _deferral["active"]
             and not backend.foreign_capture())
    if not (WGRAD_STREAM or small):
        return None
    dev = inputs[0].device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    for t in inputs:
        t.record_stream(side)
    side.deferred = bool(small and not WGRAD_STREAM)
    return side


def join(side, *outputs):
    """The current stream waits for the side work; `outputs` (allocated on the side stream) are consumed on the current one.
    In a deferral scope (small maps) the wait is left to wait_deferred()."""
    if side is None:
        return
    main = torch.cuda.current_stream(side.device)
    for t in outputs:
        if t is not None:
            t.record_stream(main)
    if getattr(side, "deferred", False):
        _deferral["pending"].add(side)
        return
    main.wait_stream(side)


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


GRAD_IN_PLACE = os.environ.get("GRIT_GRAD_IN_PLACE", "1") != "0"  # A/B knob of grad_slot
DEFER_SLOT_SHORT = os.environ.get("GRIT_DEFER_SLOT_SHORT", "1") != "0"  # deferred short-map gradients written into their bucket slots


def grad_slot(param, dtype, device):
    """Inside a gradient-bucket scope: a FRESH view of `param`'s slot in its flat gradient bucket (grit_amd/ddp.py), for a backward
    node to write the gradient into -- autograd adopts the view as .grad, the bucket pack finds it in place and skips its copy
    (~400 MB of read + write per step for the Swin weight gradients).  None when it does not apply."""
    slot = getattr(param, "_grit_grad_slot", None) if (GRAD_IN_PLACE and param is not None and _deferral["active"]) else None
    # _grit_slot_open is the bucket wrapper's word (grit_amd/ddp.py): the parameter is in the live set, its bucket has not been
    # packed yet in this backward pass, and nobody has been handed the slot before -- a second node of a weight used twice gets
    # None (a fresh tensor: autograd sums the two), a parameter outside the live set or behind a sent bucket goes the late path
    # with a tensor of its own (the late path all-reduces what .grad holds; a view of the slot there reads as "already reduced")
    if slot is None or param.grad is not None or not getattr(param, "_grit_slot_open", False):
        return None
    flat, off, n, shape = slot
    if flat.dtype != dtype or flat.device != device or (flat.data_ptr() + off * flat.element_size()) % 16:
        return None
    param._grit_slot_open = False
    return flat[off:off + n].view(shape)


def slab_sum(partial, out_dtype, slabs=None, out=None):
    """f32 partial sums [groups, slabs_allocated, n...] (contiguous) -> [groups, n...] in out_dtype: the sum over the first
    `slabs` slabs of every group and the dtype cast in one launch (grit_slab_sum).  out: where to (contiguous, same shape)."""
    slabs = partial.shape[1] if slabs is None else slabs
    tail = partial.shape[2:]
    n = 1
    for d in tail:
        n *= d
    if out is None:
        out = torch.empty((partial.shape[0],) + tuple(tail), dtype=out_dtype, device=partial.device)
    with _lib.device_guard(partial.device):
        st = _lib.load().grit_slab_sum(ctypes.c_void_p(partial.data_ptr()), partial.shape[0], partial.stride(0), slabs, n,
                                       ctypes.c_void_p(out.data_ptr()), int(out_dtype == torch.bfloat16), _lib.current_stream_ptr())
    _lib.check(st, "grit_slab_sum")
    return out


class SlabGroup(object):
    """The slab sums a backward node owes, reduced by ONE grit_slab_sum_grouped launch at the end of the node instead of one
    dependent ~6 us launch each (split-M weight gradients, bias-gradient column sums, LayerNorm dgamma / dbeta ...).
    `add` allocates and returns the output tensor right away; it holds the result only after `run()`."""

    ENABLED = os.environ.get("GRIT_SLAB_GROUP", "1") != "0"  # A/B knob: 0 = one grit_slab_sum launch per job, as before

    def __init__(self):
        self.jobs, self.keep = [], []

    def add(self, partial, out_dtype, slabs=None, out=None, extra=None):
        """partial f32 [groups, slabs_allocated, n...] contiguous -> out [groups, n...] (sum over the first `slabs` slabs);
        out: an existing contiguous tensor of that shape and dtype to write to; extra: f32 [groups, n...] contiguous, one more term of
        every sum (grit_slab_job.extra)."""
        if not SlabGroup.ENABLED:
            res = slab_sum(partial, out_dtype, slabs, out)
            return res if extra is None else res.add_(extra.view(res.shape).to(res.dtype))
        slabs = partial.shape[1] if slabs is None else slabs
        tail = tuple(partial.shape[2:])
        n = 1
        for d in tail:
            n *= d
        if out is None:
            out = torch.empty((partial.shape[0],) + tail, dtype=out_dtype, device=partial.device)
        if extra is not None:
            assert extra.dtype == torch.float32 and extra.is_contiguous() and extra.numel() == partial.shape[0] * n and extra.data_ptr() % 16 == 0
            self.keep.append(extra)
        self.jobs.append((partial.data_ptr(), partial.stride(0), partial.shape[0], slabs, n, out.data_ptr(),
                          int(out_dtype == torch.bfloat16), 0 if extra is None else extra.data_ptr()))
        self.keep.append(partial)  # the partials must outlive the launch; outputs are owned by the caller
        self.device = partial.device
        return out

    def add_raw(self, partial, groups, group_stride, slabs, n, out_ptr, out_is_bf16):
        """Same, for a partial block inside a larger workspace and an output that already exists (raw device address)."""
        self.jobs.append((partial.data_ptr(), group_stride, groups, slabs, n, out_ptr, int(out_is_bf16)))
        self.keep.append(partial)
        self.device = partial.device

    def run(self):
        if not self.jobs:
            return
        lib = _lib.load()
        with _lib.device_guard(self.device):
            for i in range(0, len(self.jobs), _lib.SLAB_GROUP_MAX):
                chunk = self.jobs[i:i + _lib.SLAB_GROUP_MAX]
                table = (_lib.SlabJob * len(chunk))(*[_lib.SlabJob(*j) for j in chunk])
                st = lib.grit_slab_sum_grouped(table, len(chunk), _lib.current_stream_ptr())
                _lib.check(st, "grit_slab_sum_grouped")
        self.jobs, self.keep = [], []


def column_sum(x2d, out_dtype=torch.float32, group=None, extra=None):
    """[M, N] (bf16 / f32, contiguous, N % 8 == 0) -> [N] in out_dtype (f32 accumulation).  With `group` (a SlabGroup) the
    second stage is left to the group's launch: the returned tensor is filled by group.run()."""
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
    if group is not None:
        return group.add(partial.unsqueeze(0), out_dtype, extra=extra)[0]
    res = slab_sum(partial.unsqueeze(0), out_dtype)[0]
    return res if extra is None else res.add_(extra.to(res.dtype))


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


# GRIT_WGRAD_SMALL=1: weight + bias gradient of short maps (decoders, grid net) from ONE own launch, grit_wgrad_small, instead
# of the library's transposed GEMM + column-sum kernel.  Stand-alone the kernel wins (6-25 us against 25-31 + 13 us,
# tools/bench_small_wgrad.py); INSIDE the training step it averages 31 us per call -- operands cold in HBM, 3 workgroups per CU --
# and the step gets 0.9-1.5 ms SLOWER (65.2 -> 66.1 / 66.8 ms, two alternating passes on one box; a one-split variant that writes
# finished bf16 gradients without a reduction launch: 63.5 -> 64.8 ms): default off, profiles/r03/negative_results.txt.
WGRAD_SMALL = os.environ.get("GRIT_WGRAD_SMALL", "0") == "1"
WGRAD_SMALL_MAX_ROWS = 16384


def small_weight_bias_grad(dy2, x2, need_db, out_dtype, group=None):
    """(dW [N, K], db [N] or None) of a Linear on a short map from ONE grit_wgrad_small launch (split-M partials of both) plus
    the reduction -- the caller's SlabGroup when given -- or None when the kernel does not cover the problem."""
    M, N = dy2.shape
    K = x2.shape[1]
    if not (WGRAD_SMALL and dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and M < WGRAD_SMALL_MAX_ROWS
            and N % 64 == 0 and K % 64 == 0 and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0
            and x2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0):
        return None
    lib = _lib.load()
    S = lib.grit_wgrad_small_splits(M, N, K)
    if S <= 0:
        return None
    direct = S == 1 and out_dtype == torch.bfloat16  # the kernel rounds once and writes the finished gradients
    if direct:
        wpart = torch.empty((N, K), dtype=torch.bfloat16, device=dy2.device)
        bpart = torch.empty((N,), dtype=torch.bfloat16, device=dy2.device) if need_db else None
    elif S == 1:
        return None  # f32 gradients of a short map: the library GEMM (parity path, not a throughput one)
    else:
        wpart = torch.empty((1, S, N, K), dtype=torch.float32, device=dy2.device)
        bpart = torch.empty((1, S, N), dtype=torch.float32, device=dy2.device) if need_db else None
    with _lib.device_guard(dy2.device), timed("gemm_own", flops=2.0 * M * N * K, kernel="wgrad_small", bytes=2.0 * M * (N + K) + 4.0 * S * N * K):
        st = lib.grit_wgrad_small(ctypes.c_void_p(dy2.data_ptr()), dy2.stride(0), ctypes.c_void_p(x2.data_ptr()), x2.stride(0), M, N, K,
                                  S, ctypes.c_void_p(wpart.data_ptr()), ctypes.c_void_p(bpart.data_ptr()) if need_db else None,
                                  _lib.current_stream_ptr())
    _lib.check(st, "grit_wgrad_small")
    if direct:
        return wpart, bpart
    own = group is None
    if own:
        group = SlabGroup()
    dw = group.add(wpart, out_dtype)[0]
    db = group.add(bpart, out_dtype)[0] if need_db else None
    if own:
        group.run()
    return dw, db


# GRIT_WGRAD_TN (default 1): weight gradients of the long token maps by grit_wgrad_tn (grit_amd/csrc/wgrad_tn.hip: 256 x 256 tiles,
# transposing LDS reads, one workgroup per CU) instead of the library's batched GEMM over 16 row slices -- 1.02-1.19 PFLOP/s
# against 0.76-0.99 on the Swin shapes (profiles/r03/wgrad_tn.txt).  The slices' fp32 partials are summed by the grouped slab sum as
# before.
WGRAD_TN = os.environ.get("GRIT_WGRAD_TN", "1") != "0"
# GRIT_WGRAD_TN_GROUPED (default 1): the deferred short-map weight gradients whose shapes fit (N, K multiples of 256, M of 32) run
# through the same kernel in one grouped launch (grit_wgrad_tn_grouped), their bias gradients through grit_colsum_grouped
WGRAD_TN_GROUPED = os.environ.get("GRIT_WGRAD_TN_GROUPED", "1") != "0"
# GRIT_WGRAD_TN_BIAS (default 1): the bias gradient of a long-map Linear (qkv) as a by-product of its weight-gradient launch
WGRAD_TN_BIAS = os.environ.get("GRIT_WGRAD_TN_BIAS", "1") != "0"


# GRIT_WGRAD_ROW_SKIP (default 1): the long-map weight gradients do not load the rows of samples that drop path removed from the branch
# (exact zeros in dY); the row slices share the live rows equally (grit_wgrad_tn_rows, wgrad_tn.hip).  Also read by the library itself.
WGRAD_ROW_SKIP = os.environ.get("GRIT_WGRAD_ROW_SKIP", "1") != "0"


def _rows_arg(row_scale, M):
    """(pointer or None, rows per sample) of a (factors, rows_per_sample) pair for an M-row problem; (None, 0) when it does not apply."""
    if not WGRAD_ROW_SKIP or row_scale is None:
        return None, 0
    scale, per = row_scale
    if (scale is None or per <= 0 or not scale.is_cuda or scale.dtype != torch.float32 or not scale.is_contiguous()
            or scale.numel() * per != M):
        return None, 0
    return scale.data_ptr(), int(per)


def long_weight_grad_partials(dy2, x2, need_db=False, row_scale=None):
    """fp32 partials [S, N, K] of dW = dy2^T x2 from the own kernel, or None where it does not apply (the library path runs).
    need_db: returns (partials, [S, N] fp32 column sums of dy2 per slice) -- the bias gradient as a by-product of the same launch."""
    if not (WGRAD_TN and dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16):
        return None
    M, N = dy2.shape
    K = x2.shape[1]
    if M < 8192 or dy2.stride(1) != 1 or x2.stride(1) != 1:  # shorter maps: the deferred grouped kernel / the library
        return None
    lib = _lib.load()
    S = lib.grit_wgrad_tn_splits(M, N, K)
    if S <= 0 or dy2.stride(0) % 8 or x2.stride(0) % 8 or dy2.data_ptr() % 16 or x2.data_ptr() % 16:
        return None
    part = torch.empty((S, N, K), dtype=torch.float32, device=dy2.device)
    bpart = torch.empty((S, N), dtype=torch.float32, device=dy2.device) if need_db else None
    with _lib.device_guard(dy2.device), timed("gemm_own", flops=2.0 * M * N * K, kernel="wgrad_tn", bytes=2.0 * M * (N + K) + 4.0 * S * N * K):
        rs, per = _rows_arg(row_scale, M)
        st = lib.grit_wgrad_tn_rows(ctypes.c_void_p(dy2.data_ptr()), dy2.stride(0), ctypes.c_void_p(x2.data_ptr()), x2.stride(0), M, N, K, S,
                                    ctypes.c_void_p(part.data_ptr()), ctypes.c_void_p(bpart.data_ptr()) if need_db else None,
                                    ctypes.c_void_p(rs) if rs else None, per, _lib.current_stream_ptr())
    _lib.check(st, "grit_wgrad_tn_rows")
    return (part, bpart) if need_db else part


WGRAD_TN_PAIR = os.environ.get("GRIT_WGRAD_TN_PAIR", "1") != "0"  # the two weight gradients of a Swin Mlp as one grouped launch


def tn_slices(M, want):
    """Row slices the long-map kernel accepts for an M-row problem when `want` are asked for: every slice a whole number of
    64-row (M % 64 == 0) or 32-row steps and none of them empty (wgrad_tn.hip tn_fill rejects anything else: 58 368 rows = 1 824 steps cut 64 ways are
    29-step slices, i.e. 63 of them).  Same normalisation as grit_wgrad_tn_splits."""
    steps = max(1, M // (64 if M % 64 == 0 else 32))  # 64-row steps (the four-wave kernel's) wherever the row count allows
    want = max(1, min(int(want), steps))
    per = -(-steps // want)
    return -(-steps // per)


def long_weight_grad_with_parked(dy2, x2, group, weight, row_scale=None):
    """(partials [S, N, K], bias column sums [S, N]) of this node's long-map Linear like long_weight_grad_partials(.., True), from a
    grouped launch that also computes the parked weight gradients (park_weight_grad_for_partner); their slice sums join `group`.
    None when nothing is parked or it does not apply (the parked jobs then stay for the scope's flush)."""
    parked = _deferral["parked"]
    if not (parked and WGRAD_PARK and WGRAD_TN_BIAS and getattr(weight, "_grit_wgrad_pickup", False) and group is not None
            and dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and SlabGroup.ENABLED):
        return None
    M, N = dy2.shape
    K = x2.shape[1]
    lib = _lib.load()
    if (M < 8192 or dy2.stride(1) != 1 or x2.stride(1) != 1 or dy2.stride(0) % 8 or x2.stride(0) % 8 or dy2.data_ptr() % 16
            or x2.data_ptr() % 16 or lib.grit_wgrad_tn_group_ok(M, N, K) != 1 or len(parked) + 1 > _lib.WGRAD_GROUP_MAX
            or any(j[0].device != dy2.device for j in parked)):
        return None
    tiles = (N // 256) * (K // 256) + sum((j[7] // 256) * (j[8] // 256) for j in parked)
    if tiles > 256:
        return None
    want = min([max(1, 256 // tiles), (M // 32) // 16 or 1] + [(j[6] // 32) // 16 or 1 for j in parked])
    S = tn_slices(M, want)  # per problem: the slice count the kernel's contract admits for ITS row count
    _deferral["parked"] = []  # (from here on nothing may fail softly: the parked jobs are this launch's)
    part = torch.empty((S, N, K), dtype=torch.float32, device=dy2.device)
    bpart = torch.empty((S, N), dtype=torch.float32, device=dy2.device)
    table = (_lib.WgradJob * (len(parked) + 1))()
    table[0] = _lib.WgradJob(dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), M, N, K, S, part.data_ptr(), bpart.data_ptr(),
                             *_rows_arg(row_scale, M))
    flops = 2.0 * M * N * K
    nbytes = 2.0 * M * (N + K) + 4.0 * S * N * K
    for t, (pdy, px, pw_param, _, pw, _, pM, pN, pK, prs) in enumerate(parked):
        pS = tn_slices(pM, want)
        work = torch.empty(pS * pN * pK, dtype=torch.float32, device=dy2.device)
        table[t + 1] = _lib.WgradJob(pdy.data_ptr(), pdy.stride(0), px.data_ptr(), px.stride(0), pM, pN, pK, pS, work.data_ptr(), None,
                                     *_rows_arg(prs, pM))
        group.add_raw(work, 1, 0, pS, pN * pK, pw, True)
        _deferral["unverified"].append((pw_param, pw, "parked weight gradient [%d, %d]" % (pN, pK)))
        flops += 2.0 * pM * pN * pK
        nbytes += 2.0 * pM * (pN + pK) + 4.0 * pS * pN * pK
    with _lib.device_guard(dy2.device), timed("gemm_own", flops=flops, kernel="wgrad_tn", bytes=nbytes):
        st = lib.grit_wgrad_tn_grouped(table, len(parked) + 1, _lib.current_stream_ptr())
    _lib.check(st, "grit_wgrad_tn_grouped")
    return part, bpart


def long_weight_grads_together(pairs, row_scale=None):
    """[fp32 partials [S_j, N_j, K_j]] of dW_j = dy_j^T x_j for several long-map problems of ONE backward node from one grouped launch
    of the long-map kernel -- together their tiles fill the chip with fewer row slices each (fc1 + fc2 of a Swin Mlp: 16 + 16 tiles,
    8 slices of 200 steps instead of 16 of 100: half the fp32 slices to write and to sum).  None when it does not apply."""
    if not (WGRAD_TN and WGRAD_TN_PAIR and len(pairs) > 1):
        return None
    lib = _lib.load()
    tiles = 0
    for dy2, x2 in pairs:
        M, N = dy2.shape
        K = x2.shape[1]
        if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and M >= 8192 and dy2.stride(1) == 1
                and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0
                and x2.data_ptr() % 16 == 0 and lib.grit_wgrad_tn_group_ok(M, N, K) == 1):
            return None
        tiles += (N // 256) * (K // 256)
    if tiles > 256:
        return None
    want = max(1, 256 // tiles)
    want = min([want] + [(dy2.shape[0] // 32) // 16 or 1 for dy2, _ in pairs])
    table = (_lib.WgradJob * len(pairs))()
    parts = []
    for t, (dy2, x2) in enumerate(pairs):
        M, N = dy2.shape
        K = x2.shape[1]
        S = tn_slices(M, want)
        part = torch.empty((S, N, K), dtype=torch.float32, device=dy2.device)
        parts.append(part)
        table[t] = _lib.WgradJob(dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), M, N, K, S, part.data_ptr(), None,
                                 *_rows_arg(row_scale, M))
    with _lib.device_guard(pairs[0][0].device), timed("gemm_own", flops=2.0 * sum(d.shape[0] * d.shape[1] * x.shape[1] for d, x in pairs), kernel="wgrad_tn",
                                                      bytes=sum(2.0 * d.shape[0] * (d.shape[1] + x.shape[1]) + 4.0 * p.numel() for (d, x), p in zip(pairs, parts))):
        st = lib.grit_wgrad_tn_grouped(table, len(pairs), _lib.current_stream_ptr())
    _lib.check(st, "grit_wgrad_tn_grouped")
    return parts


def weight_grad(dy2, x2, group=None, param=None, row_scale=None):
    """dW [N, K] = dy2^T [N, M] @ x2 [M, K], split over M into one batched GEMM with fp32 partial sums.  With `group` (a
    SlabGroup) the sum over the partials is left to the group's launch.  param: the weight this is the gradient of -- inside a
    gradient-bucket scope the sum is then written straight into the parameter's bucket slot (grad_slot)."""
    small = small_weight_bias_grad(dy2, x2, False, dy2.dtype, group)
    if small is not None:
        return small[0]
    M, N = dy2.shape
    own = long_weight_grad_partials(dy2, x2, row_scale=row_scale)
    if own is not None:
        slot = grad_slot(param, dy2.dtype, dy2.device)
        out = None if slot is None else slot.view(1, N, x2.shape[1])
        if group is not None:
            return group.add(own.unsqueeze(0), dy2.dtype, out=out)[0]
        return slab_sum(own.unsqueeze(0), dy2.dtype, out=out)[0]
    S = split_k(M) if (dy2.is_cuda and dy2.dtype == torch.bfloat16) else 1
    with timed("gemm_lib", **gemm_work(M, N, x2.shape[1], partial_f32=0.0)):
        if S == 1:
            return torch.mm(dy2.t(), x2)
        part = torch.bmm(dy2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, x2.shape[1]), out_dtype=torch.float32)
    if group is not None:
        return group.add(part.unsqueeze(0), dy2.dtype)[0]
    return slab_sum(part.unsqueeze(0), dy2.dtype)[0]


def weight_bias_grad(dy2, x2, group, need_w, need_b, weight, row_scale=None, bias_extra=None):
    """(dW, db) of a Linear from dy2 [M, N], x2 [M, K]; either may be None when not wanted.  Long maps with both wanted: ONE launch of
    the own kernel yields the weight-gradient slices and, as a by-product, the bias gradient's column sums (no pass over dy2 of its
    own); otherwise weight_grad / column_sum.  group: the node's SlabGroup (the sums are left to its launch)."""
    pair = long_weight_grad_with_parked(dy2, x2, group, weight, row_scale) if (need_w and need_b) else None
    if pair is None:
        pair = long_weight_grad_partials(dy2, x2, True, row_scale) if (WGRAD_TN_BIAS and need_w and need_b and group is not None) else None
    if pair is not None:
        slot = grad_slot(weight, dy2.dtype, dy2.device)
        dw = group.add(pair[0].unsqueeze(0), dy2.dtype, out=None if slot is None else slot.view(1, dy2.shape[1], x2.shape[1]))[0]
        return dw, group.add(pair[1].unsqueeze(0), weight.dtype, extra=bias_extra)[0]
    dw = weight_grad(dy2, x2, group, param=weight, row_scale=row_scale) if need_w else None
    db = column_sum(dy2 if dy2.is_contiguous() else dy2.contiguous(), weight.dtype, group, extra=bias_extra) if need_b else None
    return dw, db


def _own_linear(x, weight, bias):
    """F.linear on the own long-map kernel (grit_amd/ops/gemm.py long_linear) or None."""
    from grit_amd.ops import gemm as _gemm
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.numel() // x.shape[-1] >= min(512, _gemm.SHORT_MIN_ROWS)):
        return None
    x2 = x.reshape(-1, x.shape[-1])
    y = _gemm.long_linear(x2 if x2.is_contiguous() else x2.contiguous(), weight, bias)
    return None if y is None else y.view(x.shape[:-1] + (weight.shape[0],))


def own_or_library_linear(x, weight, bias):
    """F.linear without autograd bookkeeping of its own: the own kernels where the policies take the shape, else the library."""
    y = _own_linear(x, weight, bias) if (backend.override() is None and not torch.is_autocast_enabled()) else None
    return y if y is not None else F.linear(x, weight, bias)


def _own_input_grad(dy2, weight, shape):
    if not (dy2.is_cuda and dy2.dtype == torch.bfloat16 and dy2.shape[0] >= 512):
        return None
    from grit_amd.ops import gemm as _gemm
    dx = _gemm.long_input_grad(dy2, weight)
    return None if dx is None else dx.view(shape)


def leave_bias_extra(bias, term):
    """Called from the backward of a node that uses `bias` (a Linear's bias parameter, tagged by linear() on the Linear's result) a
    second time: `term` (f32, contiguous, bias-shaped, 16-byte aligned) is its gradient contribution.  The Linear's own backward --
    which runs later in the same pass -- adds it inside its bias-gradient reduction; the caller returns None for that input."""
    bias._grit_bias_extra = term


def take_bias_extra(bias):
    term = getattr(bias, "_grit_bias_extra", None) if bias is not None else None
    if term is not None:
        del bias._grit_bias_extra
    return term


class _LinearFn(Function):

    @staticmethod
    def forward(ctx, x, weight, bias, single_use=False, row_scale=None, relu=None):
        ctx.has_bias = bias is not None
        ctx.relu = None
        if relu is not None:
            # (p, device seed, holder): dropout(relu(x W^T + b)) in the GEMM's epilogue (linear_relu_dropout checked that the short-map tile
            # takes the shape); the backward of ReLU + dropout is applied by the consumer's input-gradient GEMM when it said so in the
            # holder (grit_amd/ops/layer_norm.py), else here, from the saved output
            from grit_amd.ops import gemm as _gemm
            p, seed_dev, holder = relu
            x2 = x.reshape(-1, x.shape[-1])
            y = _gemm.gemm_nt_relu(x2 if x2.is_contiguous() else x2.contiguous(), weight, _gemm.BIAS_RELU_DROP, bias=bias, p=p,
                                   seed_dev=seed_dev).view(x.shape[:-1] + (weight.shape[0],))
            ctx.relu = (p, holder, seed_dev is not None)
            ctx.save_for_backward(*((x, weight, y) + ((seed_dev,) if seed_dev is not None else ())))
        else:
            ctx.save_for_backward(x, weight)
        ctx.single_use = single_use
        # drop path: (per-sample factors [B] f32, rows per sample) of the branch this Linear feeds -- the caller's promise that the rows
        # of a sample with factor 0 come back as exact zeros in dy (wgrad_tn.hip then skips them in the weight gradient)
        ctx.row_scale = row_scale
        ctx.bias_param = bias if single_use else None  # the parameter itself: the deferred path checks its .grad
        ctx.weight_param = weight if single_use else None
        ctx.weight_obj = weight  # the tensor object the forward was called with: transposed copies are attached to IT
        ctx.bias_obj = bias      # ... and a gradient term another node leaves for this bias (take_bias_extra)
        if relu is not None:
            return y
        own = _own_linear(x, weight, bias)
        if own is not None:
            return own
        with timed("gemm_lib", **gemm_work(x.numel() // x.shape[-1], weight.shape[0], weight.shape[1])):
            return F.linear(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        x, weight = saved[0], saved[1]
        if ctx.relu is not None and not ctx.relu[1].get("fused"):
            from grit_amd.ops.glue import relu_dropout_backward
            dy = relu_dropout_backward(saved[2], dy, ctx.relu[0], saved[3] if ctx.relu[2] else None)  # (no consumer took it over)
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        dx = dw = db = None
        if ctx.row_scale is not None:
            backend.check_dropped_rows(dy2, ctx.row_scale[0], "Linear backward (row_scale)")
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        # a second gradient term of the bias that another node of this pass left for this one (window attention: the q / k / v rows of
        # the window-padding tokens ARE the qkv bias): it joins the bias gradient's slab sum as one more term -- no cast, no add launch
        extra = take_bias_extra(ctx.bias_obj) if need_b else None
        deferred = None if extra is not None else \
            defer_weight_bias_grad(dy2, x2, weight, ctx.bias_param, ctx.needs_input_grad[1], need_b, ctx.single_use)
        if deferred is not None:  # short map inside a gradient-bucket scope: dW / db come from the scope's grouped launch
            dw, db = deferred
            if ctx.needs_input_grad[0]:
                dx = _own_input_grad(dy2, ctx.weight_obj, x.shape)
                if dx is None:
                    with timed("gemm_lib", **gemm_work(dy2.shape[0], weight.shape[1], weight.shape[0])):
                        dx = torch.mm(dy2, weight).view(x.shape)
            return dx, dw, db, None, None, None
        side = fork(dy2, x2, rows=dy2.shape[0], single_use=ctx.single_use) \
            if (ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or need_b)) else None
        group = SlabGroup() if dy2.is_cuda else None  # dW's and db's partial sums: one reduction launch
        with on_stream(side):
            both = small_weight_bias_grad(dy2, x2, need_b, weight.dtype, group) if (ctx.needs_input_grad[1] and extra is None) else None
            if both is not None:  # short map: dW and db partials from one launch
                dw, db = both
            else:
                dw, db = weight_bias_grad(dy2, x2, group, ctx.needs_input_grad[1], need_b, weight, row_scale=ctx.row_scale,
                                          bias_extra=extra)
            if side is None:
                finish_group(group, ctx.single_use, [(ctx.weight_param, dw), (ctx.bias_param, db)])
            elif group is not None:
                group.run()
        if ctx.needs_input_grad[0]:
            dx = _own_input_grad(dy2, ctx.weight_obj, x.shape)
            if dx is None:
                with timed("gemm_lib", **gemm_work(dy2.shape[0], weight.shape[1], weight.shape[0])):
                    dx = torch.mm(dy2, weight).view(x.shape)
        join(side, dw, db)
        return dx, dw, db, None, None, None


# GRIT_FFN_RELU_EPILOGUE (default 1, round 6): dropout(relu(fc1(x))) of the decoders' position-wise FFNs as the epilogue of fc1's GEMM, and
# its backward as the epilogue of fc2's input-gradient GEMM (grit_gemm_bf16_nt_relu) -- a launch less each way per FFN.  0: grit_relu_dropout_*.
RELU_EPILOGUE = os.environ.get("GRIT_FFN_RELU_EPILOGUE", "1") != "0"


def linear_relu_dropout(x, lin, p, training=True):
    """dropout(relu(lin(x)), p, training): one GEMM launch with the epilogue where the short-map tile takes the shape (the result carries a
    holder through which the consuming projection node -- linear_add_layer_norm -- announces that ITS input-gradient GEMM applies the
    backward of ReLU + dropout), else the Linear followed by grit_amd.ops.glue.relu_dropout."""
    from grit_amd.ops import gemm as _gemm
    from grit_amd.ops.glue import relu_dropout
    weight, bias = lin.weight, lin.bias
    p = float(p) if training else 0.0
    rows = x.numel() // x.shape[-1]
    fits = (RELU_EPILOGUE and backend.override() is None and x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled()
            and not backend.foreign_capture() and (x.requires_grad or weight.requires_grad) and bias is not None
            and x.dtype == weight.dtype == bias.dtype == torch.bfloat16 and 0.0 <= p < 1.0 and _gemm.OWN
            and _gemm.prefers_own_short(rows, weight.shape[0], weight.shape[1]) and weight.is_contiguous()
            and weight.data_ptr() % 16 == 0 and bias.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)
    if not fits:
        return relu_dropout(lin(x), p, training)
    seed_dev = backend.dropout_seed(x.device) if p > 0 else None
    holder = {"p": p, "seed": seed_dev, "fused": False}
    y = _LinearFn.apply(x, weight, bias, single_use_now(getattr(lin, "single_use", False)), None, (p, seed_dev, holder))
    y._grit_relu_holder = holder
    return y


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
        with timed("gemm_lib", **gemm_work(x.numel() // x.shape[-1], weights[0].shape[0] * n, weights[0].shape[1])):
            return tuple(F.linear(x, w, b) for w, b in zip(weights, biases))

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        x, weights = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        n = ctx.n
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        dx2, dws, dbs = None, [None] * n, [None] * n
        group = SlabGroup() if x2.is_cuda else None
        for l in range(n):
            if dys[l] is None:
                continue
            dy2 = dys[l].reshape(-1, dys[l].shape[-1])
            dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
            if ctx.needs_input_grad[0]:
                with timed("gemm_lib", **gemm_work(dy2.shape[0], weights[l].shape[1], weights[l].shape[0])):
                    if dx2 is None:
                        dx2 = torch.mm(dy2, weights[l])
                    else:
                        dx2.addmm_(dy2, weights[l])
            dws[l], dbs[l] = weight_bias_grad(dy2, x2, group, ctx.needs_input_grad[2 + l], ctx.needs_input_grad[2 + n + l], weights[l])
        if group is not None:
            group.run()
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


def linear(x, weight, bias, single_use=False, row_scale=None):
    """F.linear with the backward of this module.  single_use=True: the caller guarantees `weight` / `bias` receive exactly one
    gradient per backward pass (not shared between call sites), which lets small maps compute them beside the chain (fork).
    row_scale = (factors [B] float32 on the device, rows per sample) or None: drop-path factors of the branch the result feeds; the caller
    guarantees that the gradient of the result is exactly zero in the rows of samples whose factor is 0 (the weight gradient skips them)."""
    fits = (backend.override() is None and x.is_cuda and torch.is_grad_enabled()
            and not torch.is_autocast_enabled()
            and (x.requires_grad or weight.requires_grad) and x.dtype == weight.dtype
            and x.dtype in (torch.bfloat16, torch.float32) and weight.shape[0] % 8 == 0
            and x.numel() // x.shape[-1] >= MIN_ROWS)
    if not fits:
        if (backend.override() is None and x.is_cuda and not torch.is_autocast_enabled() and x.dtype == torch.bfloat16
                and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad))):
            own = _own_linear(x, weight, bias)  # frozen stage / inference: the long-map kernels where the policy prefers them
            if own is not None:
                return own
        return F.linear(x, weight, bias)
    y = _LinearFn.apply(x, weight, bias, single_use_now(single_use), row_scale, None)
    if bias is not None and bias.requires_grad:
        y._grit_bias_node = bias  # (this result's backward is _LinearFn's: it honours leave_bias_extra for this bias)
    return y


class Linear(nn.Linear):
    """Same parameters / state-dict keys as nn.Linear.  `single_use` (plain attribute, not part of the state): see linear()."""

    single_use = False

    def forward(self, input, row_scale=None):
        return linear(input, self.weight, self.bias, self.single_use, row_scale)
