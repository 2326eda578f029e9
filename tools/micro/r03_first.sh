#!/bin/bash
# round 3, first GPU call: new MSDA staged backward tests + A/B of the accumulation modes on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu > gpurun_out/r03/msda_tests.log 2>&1; echo "msda tests rc=$?"
tail -3 gpurun_out/r03/msda_tests.log
for rep in 1 2; do
  for acc in 1 0; do
    GRIT_MSDA_BWD_F32ACC=$acc timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r03/bench_acc${acc}_$rep.json 2> gpurun_out/r03/bench_acc${acc}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/bench_acc${acc}_$rep.json").read().strip().splitlines()[-1])
    print("acc=$acc rep=$rep", round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms", d["msda_backward"]["avg_launch_us"], d["config"]["msda_backward_accumulation"])
except Exception as e:
    print("acc=$acc rep=$rep failed", e)
PY
  done
done
