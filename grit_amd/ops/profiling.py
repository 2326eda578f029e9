"""Per-launch HIP-event timing hook for bench.py's analysis steps (never active in the product path: EVENTS is None).

`with timed("gemm_lib", flops=...)` records an event pair on the current stream around the enclosed launches and appends
(kind, start, end, payload) to EVENTS.  An event pair costs ~5 us of marker latency, so the GEMM hooks (hundreds per step) are
only switched on for a few extra steps AFTER the timed region."""
import torch

EVENTS = None


class timed(object):

    def __init__(self, kind, **payload):
        self.kind, self.payload = kind, payload

    def __enter__(self):
        self.on = EVENTS is not None and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on and EVENTS is not None:
            self.b.record()
            EVENTS.append((self.kind, self.a, self.b, self.payload))
        return False


def gemm_work(rows, n, k, esize=2, outputs=1, extra_in=0.0, partial_f32=0.0):
    """Payload of a GEMM launch for timed(): 2 rows n k flop, and its ALGORITHMIC bytes -- both operands once, `outputs` result maps
    of rows x n (`extra_in`: further input maps of that size, e.g. the pre-activation of the GELU' epilogue; `partial_f32`: fp32
    elements written as split partials)."""
    return {"rows": rows, "flops": 2.0 * rows * n * k,
            "bytes": float(esize) * (rows * k + n * k + (outputs + extra_in) * rows * n) + 4.0 * partial_f32}

