"""Independent branches of the training step on their own HIP streams -- while the step is being CAPTURED (round 5).

The replayed step ran its ~2 300 kernels strictly one after the other (profiles/r04/bench_bs32_steady_state.txt: wall == busy == sum of
kernel times), although the two decoders are chains of 5-15 us kernels that leave most of the chip idle.  A HIP graph keeps the
dependencies it was captured with: work captured on a forked stream becomes a parallel branch, and on replay the runtime runs
parallel branches side by side (tools/micro/r05_graph_probe.py on MI355X: two 200-kernel chains 3.33 ms serial, 2.31 ms forked, one chain
alone 1.67 ms).  Autograd runs a node's backward on the stream its forward ran on, so a branch forked in the forward pass is a parallel
branch of the backward pass as well, with the engine's own cross-stream synchronisation on every edge.

Forks are taken only under grit_amd.engine.graph_step's capture (`backend.train_capture()`): eagerly every fork costs two event
operations on a host that is already the bottleneck of that phase (measured slower in round 3).

What the bucket wrapper needs to know (grit_amd/ddp.py): gradients and deferred weight-gradient jobs are produced on whichever stream ran
the node, but packed / flushed by the hook that completes a bucket, on ITS stream.  `rendezvous()` orders that stream behind every fork
taken in this capture, `release()` orders the forks behind the pack again (their tensors may be freed -- and their memory reused on the
side stream -- once the pack has been enqueued).
"""
import os

import torch

from grit_amd.ops import backend

# Default OFF (round 5, profiles/r05/ab_fork.txt, same box, alternating passes): grid net beside the detection module 51.94 -> 52.46 ms,
# region beside grid cross-attention 51.94 -> 52.84 ms, both 52.89 ms.  Branches DO run side by side on replay (r05_graph_probe.py), but a
# graph with any parallel branch leaves the runtime's single-queue fast path: the whole 2 300-node replay pays for it, and the branches of
# this step are too short to win it back.  GRIT_STEP_FORK=1 enables the forks (tests/test_graph_step_gpu.py keeps them correct).
ENABLED = os.environ.get("GRIT_STEP_FORK", "0") == "1"
MASK = int(os.environ.get("GRIT_STEP_FORK_MASK", "255"))  # bit i: fork slot i may be taken (A/B and debugging aid)
_pool = {}      # device index -> list of side streams
_in_capture = {}  # device index -> [origin stream, side streams that joined the current capture (a stream stays in capture mode until it ends)]
_suspended = False


def active(t):
    """Fork here?  Only inside the captured training step, on a device tensor, and not while a caller has suspended forking."""
    return (ENABLED and not _suspended and t.is_cuda and backend.train_capture() and torch.cuda.is_current_stream_capturing())


class suspended(object):
    """`with suspended():` -- no forks inside (the segmented capture of a step with collectives ends and begins captures from autograd
    hooks: a fork that is open across such a cut would leave the capture)."""

    def __enter__(self):
        global _suspended
        self.prev, _suspended = _suspended, True

    def __exit__(self, *exc):
        global _suspended
        _suspended = self.prev


def begin_capture(device, origin=None):
    """Called when a capture begins on stream `origin` (default: the current one): no side stream has joined it yet."""
    device = torch.device(device)
    _in_capture[device.index or 0] = [origin if origin is not None else torch.cuda.current_stream(device)]


def end_capture(device):
    _in_capture.pop(torch.device(device).index or 0, None)


def _side(device, slot):
    idx = device.index or 0
    streams = _pool.setdefault(idx, [])
    while len(streams) <= slot:
        streams.append(torch.cuda.Stream(device=device))
    return streams[slot]


class fork(object):
    """`with fork(x, slot=0, inputs=(...)) as f:` runs the block on side stream `slot`, ordered behind everything enqueued on the current stream so
    far; `f.join(*outputs)` afterwards makes the current stream wait for it (outputs were allocated on the side stream and are consumed
    on the current one).  Inactive (see active()) it is a no-op and the block runs on the current stream."""

    def __init__(self, probe, slot=0, inputs=()):
        """inputs: tensors allocated on the current stream that the block reads (and whose backward nodes, on the side stream, read
        again): their memory must not be handed out on the current stream while side-stream work may still read it."""
        self.on = active(probe) and bool((MASK >> slot) & 1)
        self.device = probe.device if self.on else None
        self.slot = slot
        self.inputs = [t for t in (probe,) + tuple(inputs) if isinstance(t, torch.Tensor) and t.is_cuda] if self.on else []

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream(self.device)
            self.side = _side(self.device, self.slot)
            self.side.wait_stream(self.main)
            for t in self.inputs:
                t.record_stream(self.side)
            joined = _in_capture.setdefault(self.device.index or 0, [self.main])
            if self.side not in joined:
                joined.append(self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        if not self.on:
            return
        self.main.wait_stream(self.side)
        for t in outputs:
            if isinstance(t, torch.Tensor):
                t.record_stream(self.main)


def rendezvous(device=None):
    """The current stream waits for every other stream of this capture -- the origin and the side streams that joined it (no-op outside
    a captured step / without forks)."""
    if "norendezvous" in _DEBUG or not forked():
        return
    if not (backend.train_capture() and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return
    cur = torch.cuda.current_stream(device)
    for s in _in_capture.get(cur.device_index, ()):
        if s != cur:
            cur.wait_stream(s)


_DEBUG = os.environ.get("GRIT_STEP_FORK_DEBUG", "")  # bisecting aid: "norendezvous", "release"


def forked():
    """True when a side stream has joined the capture in progress (the callers' cross-stream precautions are needed)."""
    return any(len(v) > 1 for v in _in_capture.values())


def keep_for_current_stream(tensors):
    """Tensors that may have been allocated on another stream of the capture and that work just enqueued on the CURRENT stream reads:
    their memory must not be handed out again on their own stream before that work has run (record_stream; inside a capture the
    allocator then keeps the block until the capture ends)."""
    if not (forked() and backend.train_capture()):
        return
    cur = torch.cuda.current_stream()
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(cur)


def release(device=None):
    """Every other stream of this capture waits for the current stream (what it just enqueued reads tensors those streams own).
    NOT used by default: with two side streams in the capture the cross waits crash hipStreamEndCapture on ROCm 7.2 (round 5) --
    keep_for_current_stream() protects the memory instead."""
    if "release" not in _DEBUG:
        return
    if not (backend.train_capture() and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return
    cur = torch.cuda.current_stream(device)
    for s in _in_capture.get(cur.device_index, ()):
        if s != cur:
            s.wait_stream(cur)
