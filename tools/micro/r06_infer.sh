#!/bin/bash
# inference detector with the residual epilogue in the no-grad path: node test, per-kernel profile, config-5 token hash
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "residual" 2>&1 | tail -3
bash tools/micro/prof_detector.sh 2>&1 | head -14
for r in 1 0; do
GRIT_GEMM_RESIDUAL=$r timeout 600 python - <<'PY'
import json, os, torch, bench
from grit_amd.config import default_config
bench._enable_tuned_gemms()
out = bench.decode_config5(torch.device("cuda:0"), default_config())
out.pop("workload")
print("GRIT_GEMM_RESIDUAL=%s" % os.environ["GRIT_GEMM_RESIDUAL"], json.dumps(out))
PY
done 2>&1 | grep GRIT_GEMM | tee gpurun_out/r06/infer_residual.txt
