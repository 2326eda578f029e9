"""Round-5 debugging aid: capture the training step (a) as one graph with forked branches, (b) in segments around one-rank RCCL
collectives, on the small G8 batch, with faulthandler on and progress lines on stderr.
  python tools/micro/r05_capture_debug.py forks [mask]     mask: bit 0 = grid net fork, bit 1 = cross-attention fork
  python tools/micro/r05_capture_debug.py segments"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode == 'segments':
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", GRIT_DDP_SELF_COLLECTIVES="1")
os.environ["GRIT_GRAPH_DEBUG"] = "1"
if mode == 'forks' and len(sys.argv) > 2:
    os.environ["GRIT_STEP_FORK_MASK"] = sys.argv[2]

import torch  # noqa: E402

from tests.test_graph_step_gpu import _batches, _setup  # noqa: E402


def main():
    torch.cuda.set_device(0)
    if mode == 'segments':
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=0, world_size=1)
    from grit_amd.engine import graph_step
    from grit_amd.engine.caption_engine import train_xe_step
    a, b = _batches()
    wrapped, opts, loss_fn = _setup()
    print("eager", float(train_xe_step(wrapped, a, opts, loss_fn)), float(train_xe_step(wrapped, b, opts, loss_fn)), file=sys.stderr, flush=True)
    step = graph_step.GraphedXEStep(wrapped, opts, loss_fn, a, eager_steps=0)
    print("captured; plan", None if step.plan is None else [k for k, _ in step.plan], file=sys.stderr, flush=True)
    for i, x in enumerate((a, b, a, b, a, b)):
        loss = step(x)
        torch.cuda.synchronize()
        print("replay", i, float(loss), file=sys.stderr, flush=True)
    print("OK", mode, file=sys.stderr, flush=True)


main()
