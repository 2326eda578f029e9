#!/bin/bash
# The RCCL gradient-sync path on the one GPU of a box: one-rank nccl group, every collective issued (GRIT_BENCH_SELF_COLLECTIVES=1).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_ddp_gloo.py -q -m gpu -x > $O/ddp_gpu_tests.log 2>&1; echo "ddp gpu tests rc=$?"; tail -5 $O/ddp_gpu_tests.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-analysis --steps 40 --warmup 15 > $O/bench_$name.json 2> $O/bench_$name.err || tail -5 $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s %.1f img/s %.2f ms  sync=%s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("grad_allreduce", "")[:60]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for pass in 1 2; do
  run plain_$pass GRIT_X=0
  run selfcoll_allreduce_$pass GRIT_BENCH_SELF_COLLECTIVES=1
  run selfcoll_shard_$pass GRIT_BENCH_SELF_COLLECTIVES=1 GRIT_GRAD_SYNC=shard
done
