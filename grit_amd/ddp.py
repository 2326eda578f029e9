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
    own HIP stream, ordered after the producing kernels by an event, i.e. it overlaps the rest of backward;
  * the set of parameters that never receive a gradient is static in GRIT (fc_alpha2, dead
    Swin norms, class/bbox heads behind .detach(), ...).  It is discovered in the first iteration and
    then excluded, instead of torch DDP's per-iteration graph walk for find_unused_parameters=True;
  * optional bf16 transport halves the bytes on the wire (sum in bf16 over <= 8 ranks, master grads fp32);
  * buffers are never broadcast (the beam-search caches are registered buffers; reference passes
    broadcast_buffers=False for the same reason).  Parameters are broadcast from rank 0 once.

`finish_gradient_sync()` must be called after loss.backward() and before optimizer.step(); the engine's
train_xe_step does it.  Works on any backend (tests run it on gloo, world_size 2, CPU).
"""
import torch
import torch.distributed as dist
from torch import nn


class _Bucket(object):
    __slots__ = ('params', 'views', 'flat', 'pending', 'expected', 'work', 'wire', 'packed')

    def __init__(self, params, views, flat):
        self.params, self.views, self.flat = params, views, flat
        self.expected = len(params)
        self.pending = self.expected
        self.work = None
        self.wire = None
        self.packed = False


class BucketedDataParallel(nn.Module):

    def __init__(self, module, bucket_mb=64, process_group=None, wire_dtype=None, broadcast_parameters=True,
                 repack_unused=True, slot_align=1):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * 2**20)
        self.wire_dtype = wire_dtype
        self.repack_unused = repack_unused  # False: keep the bucket layout (others hold views into it)
        self.slot_align = slot_align  # every parameter's slot in a flat buffer starts at a multiple of this many elements
        self._seen = set()
        self._static_unused = None  # decided after the first iteration
        self._iteration = 0
        if self.world > 1 and broadcast_parameters:
            for p in module.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
        trainable = [p for p in module.parameters() if p.requires_grad]
        self._build_buckets(trainable)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in trainable]

    # ------------------------------------------------------------------ bucket layout
    def _build_buckets(self, params):
        """Reverse registration order; params are grouped by (device, dtype) and packed up to bucket_bytes."""
        self.buckets, self._where = [], {}
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
        al = self.slot_align
        for plist in groups:
            total = sum(-(-p.numel() // al) * al for p in plist)
            flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
            off, views = 0, []
            for p in plist:
                view = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                    p.grad = view
                views.append(view)
                off += -(-p.numel() // al) * al  # padding stays zero: it is reduced and optimised as zeros
            b = _Bucket(plist, views, flat)
            for p in plist:
                self._where[p] = b
            self.buckets.append(b)

    # ------------------------------------------------------------------ backward-time hooks
    def _on_grad(self, param):
        b = self._where.get(param)
        if b is None:
            return
        if self._static_unused is None:
            self._seen.add(param)
        b.pending -= 1
        if b.pending == 0:
            self._pack(b)
            self._launch(b)

    def _pack(self, b):
        """Gradients of the bucket -> flat buffer with one multi-tensor copy; .grad becomes the view."""
        if b.packed:
            return
        src, dst = [], []
        for p, view in zip(b.params, b.views):
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                src.append(p.grad)
                dst.append(view)
        if src:
            torch._foreach_copy_(dst, src)
        for p, view in zip(b.params, b.views):
            if p.grad is not None:
                p.grad = view
        b.packed = True

    def _launch(self, b):
        if self.world == 1 or b.work is not None:
            return
        if self.wire_dtype is not None and self.wire_dtype != b.flat.dtype:
            b.wire = b.flat.to(self.wire_dtype)
            b.work = dist.all_reduce(b.wire, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, group=self.group, async_op=True)

    def finish_gradient_sync(self):
        """Launch what is still pending (first iteration / unused params), wait, average."""
        for b in self.buckets:
            if b.work is None:
                self._pack(b)
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if b.wire is not None:
                    b.flat.copy_(b.wire)
                    b.wire = None
                b.flat.mul_(1.0 / self.world)
                b.work = None
            b.pending = b.expected
        if self._static_unused is None:
            # freeze the unused set and re-pack the buckets without those parameters
            used = [p for p in self.module.parameters() if p.requires_grad and p in self._seen]
            unused = [p for p in self.module.parameters() if p.requires_grad and p not in self._seen]
            self._static_unused = unused
            if unused and self.repack_unused:
                for p in unused:
                    p.grad = None
                self._build_buckets(used)
            elif unused:  # keep the layout, but a bucket is complete once its *used* parameters have arrived
                dead = set(unused)
                for b in self.buckets:
                    b.expected = b.pending = sum(1 for p in b.params if p not in dead)
        self._iteration += 1

    @property
    def unused_parameters(self):
        return list(self._static_unused or [])

    def forward(self, *args, **kwargs):
        self.release_gradients()
        return self.module(*args, **kwargs)

    def release_gradients(self):
        """Start of a step: .grad = None everywhere, so backward assigns instead of accumulating (no add kernels).
        (Gradient accumulation over several backward passes is therefore not supported by this wrapper.)"""
        for b in self.buckets:
            b.packed = False
            for p in b.params:
                p.grad = None

    def gradient_bytes(self):
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)
