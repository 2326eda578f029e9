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
