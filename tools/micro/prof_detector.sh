#!/bin/bash
# per-kernel time of the detector forward alone (bf16 weights, batch 64, inference)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cat > /tmp/det_only.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from grit_amd.config import default_config
from grit_amd.data import synthetic_batch
from grit_amd.models.caption import Transformer
from grit_amd.models.caption.detector import build_detector
from grit_amd.tuning import load_tuned_gemms
load_tuned_gemms()
cfg = default_config(); torch.manual_seed(0)
model = Transformer(build_detector(cfg), cfg).cuda().eval().to(torch.bfloat16)
batch = synthetic_batch(64, 640, 640, device="cuda", seed=1)
with torch.no_grad():
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vis = model.detector(batch['samples'])
        torch.cuda.synchronize(); print("detector ms %.2f" % ((time.perf_counter() - t0) * 1e3))
        time.sleep(0.01)
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_det -o det -- python3 /tmp/det_only.py > $R/gpurun_out/r06/det_only.log 2>&1
grep "detector ms" $R/gpurun_out/r06/det_only.log | tail -3
python3 $R/tools/decode_kernel_profile.py /tmp/prof_det | cut -c1-170 | tee $R/gpurun_out/r06/detector_kernel_stats.txt | head -45
