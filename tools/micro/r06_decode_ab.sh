#!/bin/bash
# config 5 (beam search) with the short-map tiles also on the decode steps' 64 .. 320-row Linears: GRIT_GEMM_SHORT_MIN_ROWS 512 (default) / 64
mkdir -p gpurun_out/r06
for pass in 1 2; do for r in ${VALS:-512 64}; do  # (default 64 since)
env ${VAR:-GRIT_GEMM_SHORT_MIN_ROWS}=$r timeout 600 python - <<'PY'
import json, os, torch, bench
from grit_amd.config import default_config
bench._enable_tuned_gemms()
out = bench.decode_config5(torch.device("cuda:0"), default_config())
out.pop("workload")
v = os.environ.get("VAR", "GRIT_GEMM_SHORT_MIN_ROWS")
print("GRIT_GEMM %s=%s" % (v, os.environ.get(v)), json.dumps(out))
PY
done; done 2>&1 | grep GRIT_GEMM | tee gpurun_out/r06/decode_${VAR:-short_min_rows}.txt
