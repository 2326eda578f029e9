"""Where does the stall of two GEMM-issuing streams under TunableOp come from?  MODE = none | empty | full (table contents),
RUN = pipelined (caption_stream body with the table left ON) | detonly (detector alone on a side stream)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.cuda.tunable as tunable
from grit_amd.config import default_config
from grit_amd.data import synthetic_batch
from grit_amd.models.caption import Transformer
from grit_amd.models.caption.detector import build_detector
from grit_amd import tuning
import inference_caption as IC

mode, run = os.environ.get("MODE", "full"), os.environ.get("RUN", "pipelined")
if mode != "none":
    path = tuning.TABLE
    if mode == "empty":
        path = "/tmp/empty_table.csv"
        open(path, "w").write("".join(l for l in open(tuning.TABLE) if l.startswith("Validator")))
    print("table loaded:", tuning.load_tuned_gemms(path))
cfg = default_config()
torch.manual_seed(0)
model = Transformer(build_detector(cfg), cfg).cuda().eval().to(torch.bfloat16)
batch = synthetic_batch(64, 640, 640, device="cuda", seed=1)
dev = torch.device("cuda", 0)
with torch.no_grad():
    if run == "detonly":
        side = torch.cuda.Stream()
        for rnd in range(3):
            ts = []
            for i in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                with torch.cuda.stream(side):
                    model.detector(batch['samples'])
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            print(mode, run, "round", rnd, ["%.1f" % t for t in ts])
    else:
        gen = IC._caption_stream_device  # the pipelined body WITHOUT the tunable switch-off
        list(gen(model, [batch['samples']] * 3, cfg, 5, dev))
        for rnd in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h0 = time.perf_counter(); hs = []
            for out in gen(model, [batch['samples']] * 6, cfg, 5, dev):
                hs.append((time.perf_counter() - h0) * 1e3); h0 = time.perf_counter()
            torch.cuda.synchronize()
            print(mode, run, "round", rnd, "ms/batch %.1f" % ((time.perf_counter() - t0) / 6 * 1e3), "host ms between yields", ["%.0f" % h for h in hs])
