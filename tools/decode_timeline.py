import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from grit_amd.config import default_config
from grit_amd.data import synthetic_batch
from grit_amd.models.caption import Transformer
from grit_amd.models.caption.detector import build_detector
cfg = default_config(); torch.manual_seed(0)
model = Transformer(build_detector(cfg), cfg).cuda().eval().to(torch.bfloat16)
batch = synthetic_batch(64, 640, 640, device="cuda", seed=1)
with torch.no_grad():
    vis = model.detector(batch['samples']); model.cached_features = True
    for i in range(2):
        model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    # host-only enqueue time vs total
    t0 = time.perf_counter()
    out = model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("decode: host enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    model.cached_features = False
    t0 = time.perf_counter(); v = model.detector(batch['samples']); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("detector bf16: host enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
