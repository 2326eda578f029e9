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

Scope: a static-shape step -- same batch shape, same `any_padding` flag, same live parameter set.  By default on ONE rank without
collectives: with world > 1 the eager step keeps the RCCL overlap that was tested on gloo.  GRIT_STEP_GRAPH_COLLECTIVES=1
(experimental) captures the bucketed all-reduces too: run on hardware with a one-rank RCCL group (tests/test_graph_step_gpu.py with GRIT_TEST_RCCL_GRAPH=1: 6 of 7 runs passed, one unexplained failure inside a full-suite run;
profiles/r04/bench_rccl_one_rank_graph.json: 52.3 ms against 53.6 ms eager on one box), never with N > 1.  It needs the capture in
thread-local error mode -- torch's ProcessGroupNCCL watchdog thread keeps querying the events of earlier collectives, which the
default global mode forbids during a capture (hipErrorStreamCaptureUnsupported invalidates the capture and the watchdog's
exception terminates the process).  The sharded optimizer's cross-step all-gather is not capturable this way.
`GraphedXEStep.matches(batch)` says whether a batch fits; callers fall back to the eager step
for the odd batch (the last one of an epoch) or keep one graph per shape.
"""
import os

import torch

from grit_amd.ops import backend
from grit_amd.utils.misc import NestedTensor

ENABLED = os.environ.get("GRIT_STEP_GRAPH", "1") != "0"


def supported(model, optimizers):
    """The wrapper must be grit_amd.amp.Bf16Compute with the flat optimizer on a GPU, and no collective may be part of the step."""
    ddp = getattr(model, 'ddp', None)
    if ddp is None or not getattr(model, 'flat_optimizer', False):
        return False
    if ddp.collective and os.environ.get("GRIT_STEP_GRAPH_COLLECTIVES") != "1":
        return False
    return all(hasattr(optimizers[k], 'prepare_replay') for k in ('model', 'backbone'))


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
            raise ValueError("GraphedXEStep needs a Bf16Compute wrapper with FlatAdam optimizers and no collectives in the step")
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
        try:
            for o in self._opts:
                o.device_hyper = True
                o.prepare_replay()
            torch.cuda.synchronize(self.device)
            self.graph = torch.cuda.CUDAGraph()
            with backend.capturing_train_step(self.device) as seeds:
                # (a wrapper with collectives -- GRIT_STEP_GRAPH_COLLECTIVES=1, experimental -- captures in thread-local error mode: the
                # process group's watchdog thread keeps querying the events of earlier collectives, which global mode forbids during a capture)
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local" if model.ddp.collective else "global"):
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
        model._grit_step_graph_taken = True
        self.replays = 0

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
        self.graph.replay()
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
        self.graph = None
