"""One cross-entropy training step (engine/caption_engine.py train_xe_step, reference :312-350) captured in a HIP graph.

Why: the step is ~2 300 kernel launches.  The backbone's GEMMs give the host a lead that the ~650 dependent 5-20 us kernels of
the two decoders eat up again, so on a slow host the step is enqueue-bound in that phase (driver box of round 3: 57.3 ms per step
against 54.2 ms of GPU work, an empty HIP-event pair reading 11.9 us instead of 4.6).  A captured step is replayed by ONE
hipGraphLaunch: the GPU runs the kernels back to back whatever the host does.

What makes the step capturable (everything else already was -- no host read, no pageable copy inside train_xe_step):
  * inputs live in static buffers; `step(batch)` copies the batch in (device-to-device, outside the graph);
  * FlatAdam reads {lr / bias_correction1, 1 / sqrt(bias_correction2)} from device memory (grit_adam_flat_dev): the host rewrites
    the table before every replay, so schedulers and step counts advance without a re-capture;
  * the dropout-seed block of grit_amd.ops.backend is refilled INSIDE the graph from torch's graph-safe generator, like every
    torch RNG op of the step: each replay draws fresh masks;
  * the weight-derived copies (transposed fc2 weights) are rebuilt inside the graph; deferred / parked weight gradients and the
    fused decoder glue take part in the capture (backend.capturing_train_step) -- the graph's private pool owns every tensor
    whose raw address a deferred job keeps.

Scope: a static-shape step -- same batch shape, same `any_padding` flag, same live parameter set, same optimizer objects and runs.
`GraphedXEStep.matches(batch)` says whether a batch fits; callers fall back to the eager step for the odd batch (the last one of an
epoch).

With collectives (N > 1 ranks; round 5) the step is captured in SEGMENTS: the capture is cut behind every bucket's pack (from the autograd
hook that completes the bucket -- the engine's device thread, hence torch's "relaxed" capture-error mode, which is also what keeps
ProcessGroupNCCL's watchdog thread, querying the events of earlier collectives, from invalidating the capture) and once more where the
step waits for the collectives.  A replayed step is then: segment 0 (forward + backward up to the first full bucket) -> all-reduce of
bucket 0 issued EAGERLY on the process group's stream -> segment 1 -> all-reduce 1 -> ... -> wait -> last segment (both Adam steps) ->
the scalar all-reduce of the loss.  ~2 300 launches become ~7 graph launches + ~7 collective calls per step, the all-reduces overlap the
rest of backward exactly as in the eager step, and no collective is ever part of a capture (round 4's experiment captured them: one
unexplained failure in seven runs; removed).  Not capturable this way: the sharded optimizer's cross-step all-gather, wire-dtype
conversion of the buckets, per-step agreement of the live set (supported() says no; the eager step runs).
"""
import os

import torch

from grit_amd.ops import backend
from grit_amd.utils.misc import NestedTensor

ENABLED = os.environ.get("GRIT_STEP_GRAPH", "1") != "0"
# Opt-in (round 6): the segmented capture begins / ends captures from autograd-hook threads in torch's relaxed capture mode; its record is
# one RCCL rank on one GPU, two gloo ranks sharing a GPU and a 300-step soak -- no run with one RCCL rank PER GPU exists (no multi-GPU box
# was ever available).  Until one is on record the N > 1 default is the eager step, whose launches are what the eager tests cover.
SEGMENTS = os.environ.get("GRIT_STEP_GRAPH_SEGMENTS", "0") == "1"


def why_not(model, optimizers):
    """None when the step can be captured, else the reason it runs as eager launches (bench.py prints it as config.step_graph_reason).
    The wrapper must be grit_amd.amp.Bf16Compute with the flat optimizer on a GPU; with collectives only the plain bucketed all-reduce,
    and only when asked for."""
    ddp = getattr(model, 'ddp', None)
    if ddp is None or not getattr(model, 'flat_optimizer', False):
        return "not a Bf16Compute wrapper with the flat optimizer"
    if ddp.collective:  # GRIT_STEP_GRAPH_SEGMENTS=1: captured in segments around the collectives; default: eager launches with N > 1
        if not SEGMENTS:
            return "process group present and GRIT_STEP_GRAPH_SEGMENTS is not 1 (segmented capture is opt-in)"
        if getattr(model, 'shard_optimizer', False) or ddp.shard_grads:
            return "sharded gradients / optimizer: the cross-step all-gather is not part of the segmented plan"
        if ddp.wire_dtype is not None:
            return "wire-dtype conversion of the buckets is not part of the segmented plan"
        if ddp.agree_every_step or ddp.check_agreement:
            return "per-step agreement of the live parameter set needs the host"
    if not all(hasattr(optimizers[k], 'prepare_replay') for k in ('model', 'backbone')):
        return "optimizers without device-side per-step scalars (FlatAdam.prepare_replay)"
    return None


def supported(model, optimizers):
    return why_not(model, optimizers) is None


def _debug(msg):
    if os.environ.get("GRIT_GRAPH_DEBUG") == "1":
        import sys
        import threading
        sys.stderr.write("[graph_step %s] %s\n" % (threading.current_thread().name, msg))
        sys.stderr.flush()


class _Cutter(object):
    """The segments of a step captured around its collectives: `plan` is the replay order -- ('graph', CUDAGraph), ('collective', bucket),
    ('wait', None).  begin() / end() may be called from different threads (autograd hooks run on the engine's device thread)."""

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.plan, self.cur = [], None

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        _debug("segment %d begins" % sum(1 for k, _ in self.plan if k == 'graph'))

    def end(self):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # ("The CUDA Graph is empty": two cuts with nothing in between)
            _debug("segment ends ...")
            self.cur.capture_end()
            _debug("... ended")
        self.plan.append(('graph', self.cur))
        self.cur = None

    def collective(self, bucket):
        self.end()
        self.plan.append(('collective', bucket))
        self.begin()

    def wait_all(self):
        self.end()
        self.plan.append(('wait', None))
        self.begin()


def abandon_capture(model, opts, iteration=None):
    """After a capture that died part-way (inside forward, backward or the optimizers): leave nothing of the phantom step behind.
    Deferred / parked weight-gradient jobs hold raw addresses into the discarded graph pool; the bucket wrapper's books (closed
    gradient slots, packed flags, iteration count) describe a pass that never ran; the optimizers would keep reading their
    per-step scalars from the device table.  The next step then runs as an ordinary eager step."""
    from grit_amd.ops import linear as _linear
    _linear.abandon_deferred()
    _linear._deferral["active"] = False
    _linear._deferral["pending"].clear()
    ddp = model.ddp
    for b in ddp.buckets:
        b.work = b.wire = None
    ddp.release_gradients()
    if iteration is not None:
        ddp._iteration = iteration
    for o in opts:
        if hasattr(o, 'device_hyper'):
            o.device_hyper = False
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class GraphedXEStep(object):

    def __init__(self, model, optimizers, loss_fn, batch, scheduler=None, eager_steps=2):
        """Runs `eager_steps` ordinary steps on `batch` first (lazy caches, the live parameter set of the bucket wrapper, the
        caching allocator), then captures one.  Call it like train_xe_step's result: `loss = step(batch)`."""
        from grit_amd.engine.caption_engine import train_xe_step
        if not supported(model, optimizers):
            raise ValueError("GraphedXEStep needs a Bf16Compute wrapper with FlatAdam optimizers (and, with collectives, the plain bucketed all-reduce)")
        if getattr(model, '_grit_step_graph_taken', False):
            # a second capture on a wrapper whose first graph was released dies in hipStreamEndCapture (ROCm 7.2, seen in bench.py):
            # refuse here, callers stay on eager launches
            raise RuntimeError("this wrapper's training step was captured once already: one step graph per wrapper and process")
        self.model, self.optimizers, self.loss_fn, self.scheduler = model, optimizers, loss_fn, scheduler
        samples = batch['samples']
        self.any_padding = samples.any_padding
        self.images = samples.tensors.clone()
        self.mask = None if samples.mask is None else samples.mask.clone()
        self.captions = batch['captions'].clone()
        self.static = {'samples': NestedTensor(self.images, self.mask, any_padding=self.any_padding), 'captions': self.captions}
        self.device = self.images.device
        for _ in range(eager_steps):
            train_xe_step(model, self.static, optimizers, loss_fn)
        self._layout = model.ddp.layout_version
        self._opts = [optimizers['model'], optimizers['backbone']]
        iteration = model.ddp._iteration
        self.plan = None
        try:
            for o in self._opts:
                o.device_hyper = True
                o.prepare_replay()
            torch.cuda.synchronize(self.device)
            if model.ddp.collective:
                self._capture_segments(model, optimizers, loss_fn, train_xe_step)
            else:
                self.graph = torch.cuda.CUDAGraph()
                with backend.capturing_train_step(self.device) as seeds:
                    with torch.cuda.graph(self.graph):
                        seeds.begin_captured_step(self.device)
                        self.loss = train_xe_step(model, self.static, optimizers, loss_fn)
        except BaseException:
            self.graph = None
            model._grit_step_graph_taken = True  # (a second capture attempt on this wrapper is not safe on this ROCm either)
            abandon_capture(model, self._opts, iteration)
            raise
        # what the recorded launches depend on beyond the batch: THESE optimizer objects, each with the run layout (start / length per
        # launch, row of the device table of per-step scalars) it had at capture time -- FlatAdam._derive_runs bumps runs_version
        self._opt_state = tuple((id(o), o.runs_version) for o in self._opts)
        model.ddp._iteration = iteration  # (the recorded pass counted itself, and it never ran: replays count in __call__)
        model._grit_step_graph_taken = True
        self.replays = 0

    def _capture_segments(self, model, optimizers, loss_fn, train_xe_step):
        """The step with collectives: one capture per stretch between two collectives (module docstring)."""
        import gc
        ddp = model.ddp
        if not ddp.capture_ready():
            raise RuntimeError("segmented capture needs the bucket wrapper in steady state (live set agreed, no late gradients): "
                               "run eager steps first")
        cutter = _Cutter()
        gc.collect()
        torch.cuda.empty_cache()
        cur = torch.cuda.current_stream(self.device)
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(cur)
        ddp._cutter = cutter
        try:
            with backend.capturing_train_step(self.device) as seeds, torch.cuda.stream(cap):
                cutter.begin()
                try:
                    seeds.begin_captured_step(self.device)
                    self.loss = train_xe_step(model, self.static, optimizers, loss_fn, gather=False)
                except BaseException as e:
                    import sys
                    import traceback
                    sys.stderr.write("graph_step: exception inside the segmented capture: %s\n%s\n" % (repr(e)[:500], traceback.format_exc()[-3000:]))
                    sys.stderr.flush()
                    raise
                finally:
                    if cutter.cur is not None:
                        cutter.end()
        finally:
            ddp._cutter = None
            for b in ddp.buckets:
                b.cut = False
        cur.wait_stream(cap)
        self.graph = cutter.plan[0][1]  # (`graph is None` means released)
        self.plan, self._pool = cutter.plan, cutter.pool

    def _replay(self):
        if self.plan is None:
            self.graph.replay()
            return
        from grit_amd.engine.caption_engine import gather_result
        ddp = self.model.ddp
        for kind, arg in self.plan:
            if kind == 'graph':
                arg.replay()
            elif kind == 'collective':
                ddp.issue(arg)  # asynchronously on the process group's stream, behind the segment just enqueued
            else:
                for b in ddp.buckets:
                    if b.work is not None:
                        b.work.wait()  # (the current stream waits; the host does not)
                        b.work = None
        gather_result(self.loss)  # the rank average of the loss, in place on the graph's output (reference caption_engine.py:340)

    def matches(self, batch, optimizers=None):
        """Does `batch` (and, when given, the `optimizers` dict of the caller) fit the captured step?"""
        s = batch['samples']
        if optimizers is not None and tuple(id(optimizers[k]) for k in ('model', 'backbone')) != tuple(i for i, _ in self._opt_state):
            return False  # rebuilt between phases (XE -> SC): the replay would step the captured objects, not the caller's
        if self.graph is None or tuple((id(o), o.runs_version) for o in self._opts) != self._opt_state:
            return False  # load_state_dict / a changed live set re-derived the runs: the recorded (start, n) per launch are stale
        return (s.tensors.shape == self.images.shape and s.tensors.dtype == self.images.dtype and s.any_padding == self.any_padding
                and (s.mask is None) == (self.mask is None) and batch['captions'].shape == self.captions.shape
                and self.model.ddp.layout_version == self._layout)

    def matches_shapes(self, batch):
        s = batch['samples']
        return (s.tensors.shape == self.images.shape and s.tensors.dtype == self.images.dtype and s.any_padding == self.any_padding
                and (s.mask is None) == (self.mask is None) and batch['captions'].shape == self.captions.shape)

    def __call__(self, batch):
        if not self.matches(batch):
            raise ValueError("batch does not fit the captured step (shape / padding flag / live parameter set / optimizer runs changed)")
        s = batch['samples']
        if s.tensors.data_ptr() != self.images.data_ptr():
            self.images.copy_(s.tensors, non_blocking=True)
            if self.mask is not None:
                self.mask.copy_(s.mask, non_blocking=True)
            self.captions.copy_(batch['captions'], non_blocking=True)
        for o in self._opts:
            o.prepare_replay()
        self._replay()
        for o in self._opts:
            o.advance()
        self.model.ddp._iteration += 1
        self.replays += 1
        if self.scheduler is not None:
            lr = self.scheduler.step()
            assert self.optimizers['model'].param_groups[0]['lr'] == lr, "LR scheduler doesn't work properly."
        return self.loss

    def release(self):
        """Back to eager steps: the optimizers take their scalars from the launch arguments again."""
        for o in self._opts:
            o.device_hyper = False
        self.graph = self.plan = None
