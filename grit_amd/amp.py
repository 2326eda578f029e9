"""bf16 compute copies + fp32 master weights, without torch.autocast.

Why not autocast: on this model it launches ~2 200 tiny kernels per step (one weight cast per Linear in forward, one
bf16->fp32 gradient cast and one accumulate per parameter in backward: 761 tensors) and keeps LayerNorm in fp32 with
casts on both sides -- ~20 ms of a 170 ms step on MI355X (profiles/r01).  Here instead

  * the module's parameters ARE bf16 tensors, each a view into one flat bf16 buffer per gradient bucket; gradients
    are produced by autograd directly into flat bf16 buckets (grit_amd.ddp.BucketedDataParallel), which is also what
    RCCL all-reduces -- half the bytes on xGMI, no staging copy;
  * fp32 master parameters and Adam moments are laid out bucket-by-bucket the same way, and the optimizer step is ONE
    kernel per contiguous run of an optimizer's parameters (FlatAdam / grit_adam_flat): it reads the bf16 gradients as
    reduced, updates master + moments and rewrites the bf16 compute weights -- 28 B of HBM traffic per parameter, no
    widening copy of the gradients, no separate refresh of the compute copy;
  * numerically sensitive spots stay fp32 by construction in the modules (softmax / LayerNorm statistics inside the
    HIP kernels, MSDA sampling locations, vocabulary logits + log-softmax, the loss).

State dicts are exported from the masters (fp32) under the reference's key names.
"""
import ctypes
import math

import torch
from torch import nn

from grit_amd.ddp import BucketedDataParallel
from grit_amd.ops import weights_epoch

SLOT_ALIGN = 8  # elements: every parameter starts 16-byte aligned in the bf16 buffers, 32-byte in the fp32 ones


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (amsgrad off, weight decay 0) over the flat training state of Bf16Compute: one grit_adam_flat
    launch per contiguous run of this optimizer's parameters in each bucket reads the bf16 gradients as reduced,
    updates fp32 master + moments and rewrites the bf16 compute copy.  `state` / `param_groups` / `state_dict()` have
    torch.optim.Adam's layout (exp_avg / exp_avg_sq are views into the flat moment buffers).

    Step counts are kept PER PARAMETER, as torch.optim.Adam does (state['step'] advances only when the parameter has a
    gradient): a parameter that joins the live set later -- the detector when cached_features flips to False
    (train_caption.py:105-107) -- starts its bias corrections at step 1 instead of inheriting the optimizer's age.  A run
    never mixes parameters of different ages (Bf16Compute._runs splits on the count).

    With Bf16Compute(shard_optimizer=True) a run is clipped to this rank's slice of its bucket: every rank steps 1/world of
    the masters and moments, and Bf16Compute.after_optimizer_step all-gathers the rewritten bf16 compute weights."""

    def __init__(self, owner, params, lr, betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._owner = owner
        mine = {p for g in self.param_groups for p in g['params']}
        self._mine = mine
        self._steps = {p: 0 for p in mine}       # optimizer steps each parameter has taken
        self._step_tensors = {}                  # age -> the tensor the parameters of that age share as state['step']
        for p in mine:  # a NEW optimizer starts from zero moments (build_optimizers is called again when the phase changes)
            m, v = owner._moment_views[p]
            m.zero_()
            v.zero_()
        self._derive_runs()

    def _derive_runs(self):
        """(run, parameters, steps taken) for every contiguous range of live parameters of one age, clipped to this rank's
        slice when the optimizer is sharded."""
        self._runs = self._owner._runs(self._mine, key=self._steps.get)
        self._layout_version = self._owner.ddp.layout_version
        # A step captured in a HIP graph holds, per run, the run's (start, n) and the ADDRESS of its row of the device table of
        # per-step scalars: a captured step is valid for exactly one value of `runs_version` (grit_amd/engine/graph_step.py checks it),
        # and the table itself is allocated once, with room to spare, so that re-deriving the runs never frees memory a graph reads
        self.runs_version = getattr(self, 'runs_version', 0) + 1
        # parameters of this optimizer outside every run (outside the live set) that share their age -- and therefore their
        # state['step'] tensor -- with a parameter that WILL be stepped: the next step must give the two groups tensors of their own
        stepped = {id(p) for _, params, _ in self._runs for p in params}
        ages = {self._steps[p] for _, params, _ in self._runs for p in params}
        self._split_ages = any(id(p) not in stepped and self._steps[p] in ages for p in self._mine)
        self._link_state()

    def _link_state(self):
        for g in self.param_groups:
            for p in g['params']:
                m, v = self._owner._moment_views[p]
                n = self._steps[p]
                if n not in self._step_tensors:
                    self._step_tensors[n] = torch.tensor(float(n))
                self.state[p] = {'step': self._step_tensors[n], 'exp_avg': m, 'exp_avg_sq': v}

    def zero_grad(self, set_to_none=True):
        pass  # gradients live in the bf16 buckets, which backward overwrites; the masters never carry .grad

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            for p in g['params']:
                st = self.state.get(p, {})
                m, v = self._owner._moment_views[p]
                if 'exp_avg' in st:
                    m.copy_(st['exp_avg'])
                    v.copy_(st['exp_avg_sq'])
                    self._steps[p] = int(float(st['step']))
        self._step_tensors = {}
        self._derive_runs()

    def _hyper(self):
        hyper = {(g['lr'], tuple(g['betas']), g['eps'], g['weight_decay'], g['amsgrad'], g['maximize']) for g in self.param_groups}
        if len(hyper) != 1:
            raise NotImplementedError("FlatAdam: the parameter groups of one optimizer must share lr / betas / eps "
                                      "(the reference's groups differ only in the ignored weight_decay_rate)")
        lr, (b1, b2), eps, wd, amsgrad, maximize = next(iter(hyper))
        if wd != 0 or amsgrad or maximize:
            raise NotImplementedError("FlatAdam implements Adam with weight_decay = 0, amsgrad = False, maximize = False")
        return float(lr), float(b1), float(b2), float(eps)

    # ---- per-step scalars in device memory (device_hyper = True): what a step captured in a HIP graph reads ---------------
    device_hyper = False

    def prepare_replay(self):
        """Write {lr / bias_correction1, 1 / sqrt(bias_correction2)} of the step every run is ABOUT to take into the device table
        the launches read (pinned host copy -> one asynchronous H2D copy on the current stream).  grit_amd.engine.graph_step calls
        this before every replay of a captured step; step() calls it itself when it is not being captured."""
        lr, b1, b2, _ = self._hyper()
        n = max(self._HYPER_ROWS, len(self._runs))
        dev = self._runs[0][0][2].device if self._runs else torch.device('cpu')
        if getattr(self, '_hyper_host', None) is None or self._hyper_host.shape[1] < len(self._runs) or \
                (self._runs and self._hyper_dev.device != dev):
            # The staging copy is a RING of pinned tables: an asynchronous H2D copy reads the pinned memory when it EXECUTES, and a
            # host that runs ahead of the device (no host sync per step) would otherwise overwrite step k's scalars with step k+1's
            # before step k's copy has run.  A slot is rewritten only after the copy that read it last has completed (its event).
            shape = (self._HYPER_SLOTS, n, 2)
            self._hyper_host = torch.zeros(shape, dtype=torch.float32).pin_memory() if dev.type == 'cuda' else torch.zeros(shape)
            self._hyper_dev = torch.zeros((n, 2), dtype=torch.float32, device=dev)
            self._hyper_events = [None] * self._HYPER_SLOTS
            self._hyper_slot = 0
        slot = self._hyper_slot
        self._hyper_slot = (slot + 1) % self._HYPER_SLOTS
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()
        host = self._hyper_host[slot]
        for i, (_, _, age) in enumerate(self._runs):
            t = age + 1
            host[i, 0] = lr / (1.0 - b1 ** t)
            host[i, 1] = 1.0 / math.sqrt(1.0 - b2 ** t)
        self._hyper_dev.copy_(host, non_blocking=True)
        if self._hyper_dev.is_cuda:
            if self._hyper_events[slot] is None:
                self._hyper_events[slot] = torch.cuda.Event()
            self._hyper_events[slot].record()

    _HYPER_SLOTS = 8
    _HYPER_ROWS = 64  # rows of the device table (one per run; a handful in practice): re-derived runs keep the allocation

    @torch.no_grad()
    def step(self, closure=None):
        from grit_amd import lib as _lib
        from grit_amd.ops import backend
        lr, b1, b2, eps = self._hyper()
        if self._layout_version != self._owner.ddp.layout_version:  # the live parameter set changed: parameters without a
            self._derive_runs()                                       # gradient are not stepped (torch.optim.Adam skips them)
        grad_scale = 1.0 / self._owner.ddp.world  # the buckets hold the SUM over ranks: the average is folded in here
        ov = backend.override()
        kernel = getattr(ov, 'adam_flat', None) if ov is not None else None  # tests on CPU inject the torch restatement
        lib = _lib.load() if kernel is None else None
        from_device = self.device_hyper and kernel is None
        capturing = from_device and torch.cuda.is_current_stream_capturing()
        if from_device and (not capturing or getattr(self, '_hyper_host', None) is None):
            if capturing:
                raise RuntimeError("FlatAdam: call prepare_replay() once before capturing a step (the device table of the per-step "
                                   "scalars must exist before the capture begins)")
            self.prepare_replay()
        for i, ((bucket, compute, master, mom, var, start, end), params, age) in enumerate(self._runs):
            t = age + 1
            bc1, bc2s = 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)
            n = end - start
            if n <= 0:
                continue
            if kernel is not None:
                kernel(master[start:end], bucket.flat[start:end], mom[start:end], var[start:end], compute[start:end], float(lr),
                       float(b1), float(b2), float(eps), bc1, bc2s, grad_scale)
                continue
            with _lib.device_guard(master.device):
                if from_device:
                    st = lib.grit_adam_flat_dev(
                        ctypes.c_void_p(master[start:].data_ptr()), ctypes.c_void_p(bucket.flat[start:].data_ptr()),
                        int(bucket.flat.dtype == torch.bfloat16), ctypes.c_void_p(mom[start:].data_ptr()),
                        ctypes.c_void_p(var[start:].data_ptr()), ctypes.c_void_p(compute[start:].data_ptr()), n,
                        float(b1), float(b2), float(eps), grad_scale, ctypes.c_void_p(self._hyper_dev[i].data_ptr()),
                        _lib.current_stream_ptr())
                else:
                    st = lib.grit_adam_flat(
                        ctypes.c_void_p(master[start:].data_ptr()), ctypes.c_void_p(bucket.flat[start:].data_ptr()),
                        int(bucket.flat.dtype == torch.bfloat16), ctypes.c_void_p(mom[start:].data_ptr()),
                        ctypes.c_void_p(var[start:].data_ptr()), ctypes.c_void_p(compute[start:].data_ptr()), n, float(lr),
                        float(b1), float(b2), float(eps), bc1, bc2s, grad_scale, _lib.current_stream_ptr())
            _lib.check(st, "grit_adam_flat")
        if not capturing:  # a capture records the launches only; the replaying caller advances the books once per replay
            self.advance()
        return None

    def advance(self):
        """Host-side books of one optimizer step: every stepped parameter is one step older (the runs stay valid because the ages
        of a run move together), state['step'] follows, derived-weight caches are invalidated."""
        ages = {age for _, _, age in self._runs}
        for i, (run, params, age) in enumerate(self._runs):
            for p in params:
                self._steps[p] = age + 1
            self._runs[i] = (run, params, age + 1)
        if self._split_ages:  # unstepped parameters keep their age: fresh step tensors per age for everybody, once
            self._split_ages = False
            self._step_tensors = {}
            self._link_state()
        else:
            tensors = {}
            for age in ages:  # state['step'] tensors move with their parameters
                tnsr = self._step_tensors.pop(age, None)
                if tnsr is not None:
                    tnsr.fill_(float(age + 1))
                    tensors[age + 1] = tnsr
            merged = any(a in self._step_tensors for a in tensors)
            self._step_tensors.update(tensors)
            if merged:
                self._link_state()
        self._owner._masters_current = False
        weights_epoch.bump()  # the compute weights were rewritten by a raw kernel: no version counter saw it


class Bf16Compute(nn.Module):

    def __init__(self, module, bucket_mb=64, process_group=None, flat_optimizer=None, shard_optimizer=False):
        """shard_optimizer=True (needs the flat optimizer): gradients are reduce-scattered, every rank's FlatAdam steps only its
        1/world slice of each bucket's masters and moments, and after_optimizer_step() all-gathers the rewritten bf16 compute
        weights (waited for at the start of the next forward).  Masters / moments outside a rank's slice go stale until
        consolidate() -- a collective every rank must enter -- which the engine calls before checkpoints are written."""
        super().__init__()
        import torch.distributed as dist
        names = {p: n for n, p in module.named_parameters()}
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            for p in module.parameters():  # rank 0's fp32 initialisation is THE model: masters must start from it
                dist.broadcast(p.data, src=0, group=process_group)
        fp32 = {p: p.detach().clone().float() for p in module.parameters() if p.requires_grad}
        # floating tensors that are not trained (frozen stages, buffers) keep an fp32 original for checkpoints: exporting the
        # bf16 compute copy would round a frozen pretrained detector in every save
        self._frozen_fp32 = {k: v.detach().clone() for k, v in module.state_dict().items()
                             if v.is_floating_point() and v.dtype != torch.bfloat16
                             and not (k in dict(module.named_parameters()) and dict(module.named_parameters())[k].requires_grad)}
        module.to(torch.bfloat16)  # parameters and floating buffers; integer buffers untouched
        # flat Adam (grit_adam_flat) when the state lives on a GPU; on the CPU (gloo tests) torch's Adam steps the masters
        # from fp32 copies of the gradients and the compute weights are refreshed by a copy
        self.flat_optimizer = all(p.is_cuda for p in fp32) if flat_optimizer is None else flat_optimizer
        if shard_optimizer and not self.flat_optimizer:
            raise ValueError("shard_optimizer needs the flat optimizer (FlatAdam): torch.optim.Adam steps whole parameters")
        self.shard_optimizer = bool(shard_optimizer)
        self.group = process_group
        self.ddp = BucketedDataParallel(module, bucket_mb=bucket_mb, process_group=process_group, repack_unused=False,
                                        broadcast_parameters=False, slot_align=SLOT_ALIGN, average=not self.flat_optimizer,
                                        shard_grads=self.shard_optimizer)
        self._gather_work = []         # all-gathers of the compute weights in flight (shard_optimizer)
        self._masters_current = True   # False between a sharded optimizer step and consolidate()
        self.module = module
        self._masters, self._pairs, self._moment_views, self._slots = [], [], {}, []
        for b in self.ddp.buckets:
            n = b.flat.numel()
            compute_flat = torch.zeros(n, dtype=torch.bfloat16, device=b.flat.device)
            master_flat = torch.zeros(n, dtype=torch.float32, device=b.flat.device)
            if self.flat_optimizer:
                master_grad = None
                mom, var = torch.zeros_like(master_flat), torch.zeros_like(master_flat)
            else:
                master_grad = torch.zeros(n, dtype=torch.float32, device=b.flat.device)
                mom = var = None
            for p, gview in zip(b.params, b.views):
                k = p.numel()
                off = gview.storage_offset() - b.flat.storage_offset()
                m = nn.Parameter(master_flat[off:off + k].view_as(p))
                m.data.copy_(fp32[p])
                if master_grad is not None:
                    m.grad = master_grad[off:off + k].view_as(p)
                else:
                    self._moment_views[m] = (mom[off:off + k].view_as(p), var[off:off + k].view_as(p))
                compute_flat[off:off + k].view_as(p).copy_(m.data)
                p.data = compute_flat[off:off + k].view_as(p)  # the module now computes on the flat bf16 copy
                self._masters.append((names[p], m))
                self._slots.append((m, len(self._pairs), off, off + -(-k // SLOT_ALIGN) * SLOT_ALIGN, p))
            self._pairs.append((b, compute_flat, master_flat, master_grad, mom, var))
        self._master_of = {name: m for name, m in self._masters}
        self._compute_of = {name: p for p, name in names.items()}
        # `wrapped.module.load_state_dict(ckpt['state_dict'])` (reference train_caption.py:131-132, before every self-critical
        # epoch) must reach the fp32 masters, not only the bf16 compute views the module's parameters are
        module._register_load_state_dict_pre_hook(self._on_module_load)
        if self.shard_optimizer:  # also direct calls of wrapped.module(...) must not read weights whose all-gather is in flight
            module.register_forward_pre_hook(lambda mod, args: self.wait_for_weights())
        weights_epoch.bump()  # every parameter now lives in other storage

    # ------------------------------------------------------------------ what the engine calls
    def forward(self, *args, **kwargs):
        self.wait_for_weights()
        return self.ddp(*args, **kwargs)  # resets .grad to None first: backward assigns, one packed copy per bucket

    def wait_for_weights(self):
        """Sharded optimizer: the all-gathers of the compute weights issued by after_optimizer_step() must have landed before
        anything reads the weights.  On RCCL `wait()` makes the current stream wait for the collective's stream (no host
        block); the gathers were issued in the order the forward pass needs the buckets."""
        for w in self._gather_work:
            w.wait()
        self._gather_work = []

    def named_master_parameters(self):
        return list(self._masters)

    def flat_adam(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        """Optimizer factory for build_optimizers: Adam over `params` (masters, or Adam-style groups of them)."""
        return FlatAdam(self, params, lr=lr, betas=betas, eps=eps)

    def _runs(self, masters, key=None):
        """[(run, parameters, key value)]: maximal contiguous slot ranges of each bucket whose parameters all belong to
        `masters`, are live and share `key(parameter)` (FlatAdam: the number of steps taken).  run = (bucket, compute, master,
        mom, var, start, end) in elements of the bucket's flat buffers, clipped to this rank's slice under shard_optimizer."""
        runs, cur = [], None
        dead = self.ddp._dead
        for m, bi, start, end, p in self._slots:
            k = key(m) if key is not None else None
            if m in masters and p not in dead:
                if cur is not None and cur[0] == bi and cur[2] == start and cur[4] == k:
                    cur[2] = end
                    cur[3].append(m)
                else:
                    if cur is not None:
                        runs.append(cur)
                    cur = [bi, start, end, [m], k]
            elif cur is not None:
                runs.append(cur)
                cur = None
        if cur is not None:
            runs.append(cur)
        out = []
        for bi, start, end, params, k in runs:
            b, compute, master, _, mom, var = self._pairs[bi]
            if self.shard_optimizer:
                start, end = max(start, b.lo), min(end, b.hi)  # empty (end <= start) when the run lies in other ranks' slices
            out.append(((b, compute, master, mom, var, start, end), params, k))
        return out

    def finish_gradient_sync(self):
        self.ddp.finish_gradient_sync()
        if not self.flat_optimizer:
            for b, _, _, master_grad, _, _ in self._pairs:
                master_grad.copy_(b.flat)  # bf16 -> fp32, one kernel per bucket
            # torch.optim.Adam skips parameters whose .grad is None: parameters outside the live set must not be stepped
            dead = self.ddp._dead
            for m, bi, start, end, p in self._slots:
                if p in dead:
                    m.grad = None
                elif m.grad is None:
                    m.grad = self._pairs[bi][3][start:start + m.numel()].view_as(m)

    def after_optimizer_step(self):
        if not self.flat_optimizer:
            for b, compute_flat, master_flat, _, _, _ in self._pairs:
                compute_flat.copy_(master_flat)  # fp32 -> bf16
            weights_epoch.bump()  # the parameters are views of compute_flat: their version counters did not move
        elif self.shard_optimizer and self.ddp.collective:
            import torch.distributed as dist
            # each rank rewrote the compute weights of its slice: gather the others'.  In place (the input is the rank's slice
            # of the output: the layout RCCL's in-place all-gather defines), asynchronous, LAST bucket first -- buckets are
            # laid out in reverse registration order, so the last one holds the parameters the next forward touches first
            for b, compute_flat, _, _, _, _ in reversed(self._pairs):
                self._gather_work.append(dist.all_gather_into_tensor(compute_flat, compute_flat[b.lo:b.hi], group=self.group,
                                                                     async_op=True))

    def consolidate(self):
        """COLLECTIVE (every rank): bring the fp32 masters and Adam moments of all slices to every rank, so that
        master_state_dict() / optimizer.state_dict() describe the whole model.  No-op unless the optimizer is sharded."""
        self.wait_for_weights()
        if self.shard_optimizer and self.ddp.collective and not self._masters_current:
            import torch.distributed as dist
            for b, _, master_flat, _, mom, var in self._pairs:
                for buf in (master_flat, mom, var):
                    dist.all_gather_into_tensor(buf, buf[b.lo:b.hi], group=self.group)
        self._masters_current = True

    def master_state_dict(self):
        """fp32 state dict under the reference's key names: masters for trainable tensors, the kept fp32 originals for frozen
        ones, upcast copies for floating buffers that only exist in bf16 (beam caches)."""
        if self.shard_optimizer and self.ddp.collective and not self._masters_current:
            raise RuntimeError("sharded optimizer: the masters outside this rank's slice are stale -- call consolidate() on "
                               "EVERY rank first (the engine does at the end of an epoch)")
        self.wait_for_weights()
        sd = {k: (v.float() if v.is_floating_point() else v).clone() for k, v in self.module.state_dict().items()}
        for name, v in self._frozen_fp32.items():
            if name in sd and sd[name].shape == v.shape:
                sd[name] = v.detach().clone()
        for name, m in self._masters:
            sd[name] = m.detach().clone()
        return sd

    def _on_module_load(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Pre-hook of module.load_state_dict (root call only): incoming values go to the fp32 masters / frozen originals in
        full precision; the regular load then writes the same values, rounded, into the bf16 compute views."""
        if prefix != '':
            return
        weights_epoch.bump()
        with torch.no_grad():
            for name, m in self._masters:
                v = state_dict.get(name)
                if v is not None and v.shape == m.shape:
                    m.data.copy_(v)
            for name in list(self._frozen_fp32):
                v = state_dict.get(name)
                if v is not None and v.shape == self._frozen_fp32[name].shape:
                    self._frozen_fp32[name] = v.detach().to(self._frozen_fp32[name].device, torch.float32).clone()

    def load_master_state_dict(self, state_dict, strict=True):
        """Load a checkpoint made by master_state_dict() / the reference (key 'state_dict'): masters in fp32, compute copies
        refreshed from them.  Optimizer moments are not touched (load the optimizers' own state dicts for a resume)."""
        return self.module.load_state_dict(state_dict, strict=strict)

    @property
    def unused_parameters(self):
        return self.ddp.unused_parameters

    def gradient_bytes(self):
        return self.ddp.gradient_bytes()
