#!/bin/bash
# round-3 bench lines and profiles (one box): tests, default line, rocprof stats of the same command, steady state, PMC, A/Bs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03; mkdir -p $O
GRIT_TEST_SEED_GUARD=1 timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | tail -8
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default:", round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms; roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), "traffic", d["roofline"]["traffic"],
      "; msda", round(d["roofline_msda"]["frac"],3), "; decode", round(d["decode_config5"]["captions_per_sec_sequential"],1), "cap/s; cpu", round(d["cpu_baseline"]["value"],4), d["cpu_baseline"]["cores"], round(d["cpu_baseline"]["leg_seconds"],1), "s")
PY
bash tools/micro/ab_env.sh GRIT_MSDA_BWD_F32ACC 1 0
