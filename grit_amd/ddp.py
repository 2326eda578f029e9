"""Data-parallel gradient synchronisation for one process per GPU over RCCL/xGMI.

The reference wraps the model in torch DDP (train_caption.py:61: find_unused_parameters=True,
broadcast_buffers=False) and lets NCCL's ring all-reduce 25 MB buckets.  This is the MI355X-side design:

  * gradients live in a few large *flat* buffers (default 64 MiB each; 288 GB of HBM makes big, few
    buckets the right trade on point-to-point xGMI where a collective is per-link bound and each extra
    launch costs latency).  Autograd hands every parameter a fresh gradient tensor (.grad is None when
    backward starts, so no per-parameter accumulate kernel runs); when the last gradient of a bucket has
    arrived ONE multi-tensor copy packs them into the flat buffer, .grad is re-pointed at the views and the
    collective starts -- the optimizer then reads the reduced values in place;
  * buckets are filled in reverse registration order (~ reverse of forward = the order backward produces
    gradients); a post-accumulate-grad hook counts arrivals and, when a bucket is complete, launches
    `all_reduce(flat, async_op=True)` -- on the `nccl` (= RCCL) backend that runs on the process group's
    own HIP stream, ordered after the producing kernels by an event, i.e. it overlaps the rest of backward.
    The LAST bucket (the parameters whose gradients arrive last: nothing is left to overlap it with) is kept
    small (`tail_mb`, default 8 MiB), so the exposed part of the reduction is a small collective;
  * the set of parameters that receive no gradient is static in GRIT most of the time (fc_alpha2, dead
    Swin norms, class/bbox heads behind .detach(), ...) but it does change between phases of the reference's recipe
    (cached-feature epochs with the detector unused, then `model.module.cached_features = False`,
    train_caption.py:105-107).  The ranks agree on which parameters received a gradient anywhere (one tiny
    all-reduce of a flag vector: when the live set is first decided and whenever a rank sees a change; every step
    with agree_every_step=True); a bucket waits only for the parameters of the current *live* set, a gradient that
    arrives for a parameter outside it (or after its bucket was sent) is reduced separately in that one step, and
    the live set -- bucket completion counts, or the bucket layout itself -- is rebuilt whenever the agreed set
    changes.  This replaces torch DDP's per-iteration graph walk for find_unused_parameters=True and cannot
    silently drop or corrupt a gradient;
  * slots of live parameters that got no gradient in a step are zeroed, so an optimizer that steps whole flat
    runs never sees the previous step's values;
  * optional bf16 transport halves the bytes on the wire (sum in bf16 over <= 8 ranks, master grads fp32);
  * `shard_grads=True` ("prefer reduce-scatter + all-gather across all 7 xGMI links"): a bucket is REDUCE-SCATTERED
    instead of all-reduced -- rank r receives the sum of slice r (1/world of the bucket, slices aligned to `slot_align`) -- and
    only that slice of the flat buffer holds reduced values afterwards.  It is the first half of the sharded optimizer step
    of grit_amd.amp.Bf16Compute(shard_optimizer=True): each rank's FlatAdam updates the masters of its slice (Adam's HBM
    traffic and state divide by world) and the bf16 compute weights are all-gathered, the second half of what an all-reduce
    would have moved, now overlappable with the start of the next forward;
  * buffers are never broadcast (the beam-search caches are registered buffers; reference passes
    broadcast_buffers=False for the same reason).  Parameters are broadcast from rank 0 once.

`finish_gradient_sync()` must be called after loss.backward() and before optimizer.step(); the engine's
train_xe_step does it.  Works on any backend (tests run it on gloo, world_size 2 and 4, CPU).
"""
import torch
import torch.distributed as dist
from torch import nn

from grit_amd.ops import linear as _linear_ops


class _Bucket(object):
    __slots__ = ('params', 'views', 'flat', 'pending', 'expected', 'work', 'wire', 'packed', 'lo', 'hi', 'rs_out', 'cut')

    def __init__(self, params, views, flat, lo=0, hi=None):
        self.params, self.views, self.flat = params, views, flat
        self.lo, self.hi = lo, flat.numel() if hi is None else hi  # this rank's slice (the whole bucket unless sharded)
        self.rs_out = None
        self.expected = len(params)
        self.pending = self.expected
        self.work = None
        self.wire = None
        self.packed = False
        self.cut = False  # segmented capture: the capture was cut behind this bucket's pack


class BucketedDataParallel(nn.Module):

    def __init__(self, module, bucket_mb=64, process_group=None, wire_dtype=None, broadcast_parameters=True,
                 repack_unused=True, slot_align=1, tail_mb=8, average=True, agree_every_step=False, shard_grads=False):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.shard_grads = bool(shard_grads)
        # GRIT_DDP_SELF_COLLECTIVES=1 (measurement aid): with a process group of ONE rank still issue every collective of the
        # gradient sync (RCCL runs them as device copies).  Exercises the hooks, the bucket launches on the process group's
        # stream, the waits and the sharded optimizer on hardware that has a single GPU; the arithmetic is unchanged.
        import os
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get("GRIT_DDP_SELF_COLLECTIVES") == "1")
        self.bucket_bytes = int(bucket_mb * 2**20)
        self.tail_bytes = int(min(tail_mb, bucket_mb) * 2**20)
        self.wire_dtype = wire_dtype
        self.repack_unused = repack_unused  # False: keep the bucket layout (others hold views into it)
        self.slot_align = slot_align  # every parameter's slot in a flat buffer starts at a multiple of this many elements
        # average=False: the 1/world factor is left to the optimizer (FlatAdam folds it into its gradient scale)
        self.average = average
        # The used-parameter flags are all-reduced when the live set is first decided and whenever THIS rank sees a change
        # (a late gradient, a different used set).  That assumes what GRIT guarantees -- the graph does not depend on the
        # data, so every rank sees the change in the same step.  agree_every_step=True reduces the flags in every step
        # (safe for data-dependent graphs; costs a host read of the flags per step).
        self.agree_every_step = agree_every_step
        self.check_agreement = os.environ.get("GRIT_DDP_CHECK_AGREEMENT", "0") == "1"
        self._dead = set()        # agreed: parameters outside the live set (no gradient anywhere in the last step)
        self._used_now = set()    # parameters whose hook fired in the current backward pass
        self._late = set()        # ... and whose gradient is not part of a bucket reduction of this step
        self._decided = False     # the live set has been agreed at least once
        self.layout_version = 0   # bumped whenever the live set changes (FlatAdam re-derives its runs)
        self._iteration = 0
        self._cutter = None       # set by grit_amd.engine.graph_step while it captures the step in segments around the collectives
        if self.world > 1 and broadcast_parameters:
            for p in module.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
        self._params = [p for p in module.parameters() if p.requires_grad]
        self._index = {p: i for i, p in enumerate(self._params)}
        self._build_buckets(self._params)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self._params]

    # ------------------------------------------------------------------ bucket layout
    def _build_buckets(self, params):
        """Reverse registration order; params are grouped by (device, dtype) and packed up to bucket_bytes; the last
        group (gradients that arrive last) is cut down to tail_bytes."""
        self.buckets, self._where, self._view_of = [], {}, {}
        cur, cur_bytes, key = [], 0, None
        groups = []
        for p in reversed(params):
            k = (p.device, p.dtype)
            nbytes = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > self.bucket_bytes):
                groups.append(cur)
                cur, cur_bytes = [], 0
            key = k
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        if groups and len(groups[-1]) > 1:
            last, tail, tail_bytes = groups[-1], [], 0
            while len(last) > 1 and tail_bytes + last[-1].numel() * last[-1].element_size() <= self.tail_bytes:
                p = last.pop()
                tail.insert(0, p)
                tail_bytes += p.numel() * p.element_size()
            if tail and last:
                groups.append(tail)
            elif tail:
                groups[-1] = tail
        al = self.slot_align
        for plist in groups:
            total = sum(-(-p.numel() // al) * al for p in plist)
            if self.shard_grads:  # equal slices per rank, each a multiple of the slot alignment (padding is reduced as zeros)
                unit = al * self.world
                total = -(-total // unit) * unit
            flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
            off, views = 0, []
            for p in plist:
                view = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                    p.grad = view
                views.append(view)
                off += -(-p.numel() // al) * al  # padding stays zero: it is reduced and optimised as zeros
            if self.shard_grads:
                k = total // self.world
                b = _Bucket(plist, views, flat, self.rank * k, (self.rank + 1) * k)
            else:
                b = _Bucket(plist, views, flat)
            for p, view in zip(plist, views):
                self._where[p] = b
                self._view_of[p] = view
                # backward nodes that produce this gradient with a kernel of their own may write it straight into its slot
                # (grit_amd/ops/linear.py grad_slot): _pack then finds it in place and copies nothing
                p._grit_grad_slot = (flat, view.storage_offset(), p.numel(), tuple(p.shape))
            self.buckets.append(b)
        self._refresh_expected()

    def _refresh_expected(self):
        for b in self.buckets:
            b.expected = b.pending = sum(1 for p in b.params if p not in self._dead)
        self._open_slots()

    def _open_slots(self):
        """Which parameters may have their gradient written straight into the bucket slot in the coming backward pass
        (ops/linear.py grad_slot): those of the live set.  _pack closes the slots of its bucket, grad_slot closes a slot it hands
        out -- a gradient that arrives outside the bucket reductions must live in a tensor of its own (the late path)."""
        dead = self._dead
        for p in self._params:
            p._grit_slot_open = p not in dead

    # ------------------------------------------------------------------ backward-time hooks
    def _on_grad(self, param):
        self._used_now.add(param)
        b = self._where.get(param)
        if b is None or param in self._dead or b.packed:
            # outside the live set, or its bucket has already been sent: reduced on its own in finish_gradient_sync
            self._late.add(param)
            return
        b.pending -= 1
        if b.pending == 0:
            self._pack(b)
            self._launch(b)

    def _pack(self, b):
        """Gradients of the bucket -> flat buffer with one multi-tensor copy; .grad becomes the view.  Slots of live
        parameters without a gradient are zeroed (the flat buffer still holds the previous step's values there)."""
        if b.packed:
            return
        _linear_ops.wait_deferred()  # small-map weight gradients computed beside the backward chain (grit_amd/ops/linear.py)
        src, dst, stale = [], [], []
        for p, view in zip(b.params, b.views):
            if p.grad is None:
                if p not in self._dead:
                    stale.append(view)
            elif p.grad.data_ptr() != view.data_ptr():
                src.append(p.grad)
                dst.append(view)
        if src:
            torch._foreach_copy_(dst, src)
        if stale:
            torch._foreach_zero_(stale)
        for p, view in zip(b.params, b.views):
            p._grit_slot_open = False
            if p.grad is not None:
                p.grad = view
        b.packed = True

    def _launch(self, b):
        if not self.collective or b.work is not None:
            return
        if self._cutter is not None:
            # segmented capture of the step (grit_amd/engine/graph_step.py): the capture is CUT here -- the segment recorded so far ends
            # with this bucket packed, the collective itself is issued eagerly between the replays of this segment and the next
            if not b.cut:
                b.cut = True
                self._cutter.collective(b)
            return
        self.issue(b)

    def issue(self, b):
        """The bucket's collective, asynchronously on the process group's stream (ordered behind the current stream)."""
        src = b.flat
        if self.wire_dtype is not None and self.wire_dtype != b.flat.dtype:
            src = b.wire = b.flat.to(self.wire_dtype)
        if self.shard_grads:
            # out of place: the slice arrives in its own buffer and is copied over flat[lo:hi] once the collective is done
            if b.rs_out is None or b.rs_out.dtype != src.dtype or b.rs_out.numel() != b.hi - b.lo:
                b.rs_out = torch.empty(b.hi - b.lo, dtype=src.dtype, device=src.device)
            b.work = dist.reduce_scatter_tensor(b.rs_out, src, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(src, group=self.group, async_op=True)

    def capture_ready(self):
        """May the step be captured in segments around the bucket collectives?  Steady state only: the live set agreed, no late
        gradient in the last step, no per-step agreement, plain in-place all-reduce of the buckets' own dtype."""
        return (self.collective and self._decided and not self._late and not self.agree_every_step and not self.check_agreement
                and not self.shard_grads and self.wire_dtype is None)

    def _finish_captured(self):
        """finish_gradient_sync() while a segmented capture records the step: the steady-state path only (capture_ready()) -- pack and
        cut what backward left, a 'wait for every collective' cut, then the device-side tail.  The host-side books of a replayed step
        are kept by the replaying caller (GraphedXEStep.__call__)."""
        _linear_ops.end_deferral()
        if self._late:
            raise RuntimeError("segmented capture: a gradient arrived outside the bucket reductions (the live set changed)")
        for b in self.buckets:
            if b.expected == 0:
                continue
            if not b.packed:
                self._pack(b)
            self._launch(b)
        self._cutter.wait_all()
        if self.average and self.world > 1:
            for b in self.buckets:
                b.flat[b.lo:b.hi].mul_(1.0 / self.world)
        for b in self.buckets:
            b.cut = False
        self._refresh_expected()
        self._iteration += 1

    def finish_gradient_sync(self):
        """Launch what is still pending, agree on the used / late parameters, wait, reduce late gradients, average,
        and re-derive the live set if it changed."""
        if self._cutter is not None:
            return self._finish_captured()
        _linear_ops.end_deferral()
        for b in self.buckets:
            if b.expected == 0 and self._decided:
                continue  # every parameter of the bucket is outside the live set (agreed by all ranks): nothing to send
            if not b.packed:
                self._pack(b)
            if b.work is None:
                self._launch(b)
        # one flag per trainable parameter: bit 0 = received a gradient on this rank, bit 1 = not covered by a bucket
        flags = [0] * len(self._params)
        for p in self._used_now:
            flags[self._index[p]] = 1
        for p in self._late:
            flags[self._index[p]] = 3
        local_dead = {p for i, p in enumerate(self._params) if not flags[i] & 1}
        flag_work = flag_t = None
        need_agreement = bool(self.agree_every_step or not self._decided or self._late or local_dead != self._dead)
        if self.collective and self.check_agreement:
            # GRIT_DDP_CHECK_AGREEMENT=1 (debug): every rank must have come to the same decision, or the collectives below would
            # not match up (a data-dependent graph on one rank).  One extra tiny all-reduce + host read per step.
            probe = torch.tensor([int(need_agreement), -int(need_agreement)], dtype=torch.int32, device=self._params[0].device)
            dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=self.group)
            lo_hi = probe.cpu().tolist()
            if lo_hi[0] != -lo_hi[1]:
                raise RuntimeError("BucketedDataParallel: the ranks disagree on whether the used-parameter flags must be "
                                   "exchanged in this step (the graph depends on the data): construct with agree_every_step=True")
        if self.collective and need_agreement:
            flag_t = torch.tensor(flags, dtype=torch.int32, device=self._params[0].device)
            flag_work = dist.all_reduce(flag_t, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if self.shard_grads:
                    b.flat[b.lo:b.hi].copy_(b.rs_out)  # only this slice holds the sum over the ranks from here on
                elif b.wire is not None:
                    b.flat.copy_(b.wire)
                b.wire = None
                b.work = None
        if flag_work is not None:
            flag_work.wait()
            flags = flag_t.cpu().tolist()
        # gradients outside the bucket reductions of this step (a phase change of the model, first-step surprises): the
        # agreed list in index order, so every rank issues the same collectives.  What a slot already holds after the bucket
        # reduction (the contributions of the ranks that had the gradient in the bucket; zeros from the others: stale and
        # dead slots are kept zero) + the sum of the late local gradients = the full sum.
        scale = 1.0 / self.world
        for i, f in enumerate(flags):
            if not f & 2:
                continue
            p = self._params[i]
            view = self._view_of.get(p)
            in_bucket = view is not None and p.grad is not None and p.grad.data_ptr() == view.data_ptr()
            g = torch.zeros_like(p) if (p.grad is None or in_bucket) else p.grad
            if self.collective:
                dist.all_reduce(g, group=self.group)
            if view is not None:
                view.add_(g)
                p.grad = view
            else:
                p.grad = g.mul_(scale) if (self.average and self.world > 1) else g
        if self.average and self.world > 1:
            for b in self.buckets:
                b.flat[b.lo:b.hi].mul_(scale)
        # live set = parameters that received a gradient on some rank in this step
        dead = {p for i, p in enumerate(self._params) if not flags[i] & 1}
        if dead != self._dead or not self._decided:
            changed = dead != self._dead
            self._dead = dead
            self._decided = True
            if changed:
                if self.repack_unused:
                    for p in dead:
                        p.grad = None
                    self._build_buckets([p for p in self._params if p not in dead])
                else:
                    for b in self.buckets:
                        stale = [v for p, v in zip(b.params, b.views) if p in dead and p.grad is None]
                        if stale:
                            torch._foreach_zero_(stale)
                self.layout_version += 1
        self._refresh_expected()
        self._iteration += 1

    @property
    def unused_parameters(self):
        return [p for p in self._params if p in self._dead]

    def forward(self, *args, **kwargs):
        self.release_gradients()
        # from here to finish_gradient_sync() this wrapper is the only consumer of parameter gradients: nodes may leave the
        # weight gradients of small maps running on a side stream until _pack / finish_gradient_sync wait for it
        # (only a forward pass that records a graph opens the scope: after a no-grad / evaluation forward through the wrapper
        # nobody would call finish_gradient_sync, and a later backward elsewhere would queue jobs that are never flushed)
        if torch.is_grad_enabled():
            _linear_ops.begin_deferral(owner=self)
        else:
            _linear_ops.close_deferral(owner=self)
        return self.module(*args, **kwargs)

    def release_gradients(self):
        """Start of a step: .grad = None everywhere, so backward assigns instead of accumulating (no add kernels).
        (Gradient accumulation over several backward passes is therefore not supported by this wrapper.)"""
        self._used_now.clear()
        self._late.clear()
        for p in self._params:
            p.grad = None
        for b in self.buckets:
            b.packed = False
            b.pending = b.expected
        self._open_slots()

    def gradient_bytes(self):
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)
