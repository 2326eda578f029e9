"""Dispatch point of the three fused ops (MSDA core, Swin window attention, decoder attention).

Product behaviour: device tensors -> the gfx950 kernels of libgrit_hip.so; anything else raises
(GritHipError, "Not implemented on the CPU", like the reference's CPU stub
models/ops/src/cpu/ms_deform_attn_cpu.cpp:17-41).  There is NO automatic fallback.

The only other route is an explicit injection used by the test-suite, smoke() and bench.py's
cpu_baseline leg: `with use_reference_ops(impl):` swaps in an object (in practice oracle.torch_ref)
for the duration of the block, so the host-side logic (modules, beam search, engine) can be checked
on CPU against the golden vectors.  Nothing in grit_amd/ imports oracle/ or enters this context.
"""
from contextlib import contextmanager

_override = None


@contextmanager
def use_reference_ops(impl):
    global _override
    prev, _override = _override, impl
    try:
        yield impl
    finally:
        _override = prev


def override():
    return _override


class _SeedPool(object):
    """Dropout seeds in device memory, drawn from torch's device generator 256 at a time: a fused dropout node takes a
    one-element view (its kernels read the seed through the pointer, forward and backward), so a training step launches
    one tiny RNG kernel every few steps instead of one per node (54 per step)."""

    def __init__(self, block=256):
        self.block, self.buf, self.used = block, None, 0

    def take(self, device):
        import torch
        if self.buf is None or self.used >= self.block or self.buf.device != device:
            if train_capture() and torch.cuda.is_current_stream_capturing() and self.buf is not None:
                raise RuntimeError("dropout seed pool: a captured training step drew more than %d seeds" % self.block)
            with torch.inference_mode(False):
                self.buf = torch.empty(self.block, dtype=torch.int64, device=device).random_()
            self.used = 0
        seed = self.buf[self.used:self.used + 1]
        self.used += 1
        return seed

    def begin_captured_step(self, device):
        """First thing inside the capture of a training step: a block of seeds that the graph itself refills (torch's generator
        is graph-safe: every replay advances its Philox offset), so each replay of the step drops different elements."""
        import torch
        self.buf = torch.empty(self.block, dtype=torch.int64, device=device).random_()
        self.used = 0

    def end_captured_step(self):
        self.buf = None  # the next eager draw starts a block of its own: the captured one belongs to the graph


_seeds = _SeedPool()
_train_capture = False


def train_capture():
    """True while grit_amd.engine.graph_step captures a whole training step.  The ops that step aside under a capture they do not
    know (a beam-search graph replays with frozen weights; deferred weight gradients keep raw addresses) take part in this one: the
    graph owns every tensor of the step, and the weight-derived copies are rebuilt inside it on every replay."""
    return _train_capture


def foreign_capture():
    """The current stream is being captured by somebody other than the training-step graph."""
    import torch
    return torch.cuda.is_current_stream_capturing() and not _train_capture


@contextmanager
def capturing_train_step(device):
    global _train_capture
    prev, _train_capture = _train_capture, True
    try:
        yield _seeds
    finally:
        _train_capture = prev
        _seeds.end_captured_step()


def dropout_seed(device):
    return _seeds.take(device)


# GRIT_ROW_SKIP_CHECK=1 (debug): the row-skipping paths (Linear(row_scale=), grit_wgrad_tn_rows, grit_gemm_bf16_nt_rows, grit_winattn_bwd_bf16_rows)
# rest on the caller's promise that the gradient rows of samples whose drop-path factor is 0 are exact zeros.  With the knob set every
# such entry point verifies the promise on the host before it launches (one device sync per call: never in a timed or captured run).
import os as _os
ROW_SKIP_CHECK = _os.environ.get("GRIT_ROW_SKIP_CHECK", "0") == "1"


def check_dropped_rows(grad, row_scale, what):
    """grad [B * rows_per_sample, ...] or [B, ...]; row_scale [B] float32.  Raises when a sample with factor 0 has a non-zero gradient row."""
    import torch
    if not ROW_SKIP_CHECK or row_scale is None or grad is None or torch.cuda.is_current_stream_capturing():
        return
    B = row_scale.numel()
    g = grad.reshape(B, -1)
    bad = ((row_scale.reshape(B) == 0) & (g != 0).any(dim=1)).nonzero().flatten().tolist()
    if bad:
        raise RuntimeError("%s: row_scale promises zero gradient rows for samples %s, but they are not zero -- the caller does not "
                           "multiply the branch by the same factors" % (what, bad))
