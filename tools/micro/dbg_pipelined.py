"""Why is caption_stream slow with the tuned table loaded?  Per-batch wall time + graph cache state."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grit_amd.config import default_config
from grit_amd.data import synthetic_batch
from grit_amd.models.caption import Transformer
from grit_amd.models.caption.detector import build_detector
from grit_amd.tuning import load_tuned_gemms
from inference_caption import caption_stream
print("tuned table:", load_tuned_gemms())
cfg = default_config()
torch.manual_seed(0)
model = Transformer(build_detector(cfg), cfg).cuda().eval().to(torch.bfloat16)
batch = synthetic_batch(64, 640, 640, device="cuda", seed=1)
import grit_amd.models.caption.transformer as T
orig = T.Transformer._beam_search_graphed
calls = {"graphed": 0, "eager": 0}
def graphed(self, *a, **k):
    calls["graphed"] += 1
    return orig(self, *a, **k)
T.Transformer._beam_search_graphed = graphed
orig_e = T.Transformer._beam_search_eager
def eager(self, *a, **k):
    calls["eager"] += 1
    return orig_e(self, *a, **k)
T.Transformer._beam_search_eager = eager
with torch.no_grad():
    for rnd in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ts = []
        for out in caption_stream(model, [batch['samples']] * 5, cfg, 5):
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); t0 = time.perf_counter()
        print("round", rnd, ["%.1f" % (t * 1e3) for t in ts], calls, "graphs", len(model._decode_graphs))
    # untimed per-batch sync removed: the real pipelined loop
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = list(caption_stream(model, [batch['samples']] * 6, cfg, 5))
    torch.cuda.synchronize(); print("pipelined ms/batch %.1f" % ((time.perf_counter() - t0) / 6 * 1e3))
