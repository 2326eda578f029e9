#!/bin/bash
O=gpurun_out/r06; mkdir -p $O
rm -f $O/filled_step_measured.jsonl
for i in 1 2; do GRIT_TEST_MEASURE=$PWD/$O/filled_step_measured.jsonl timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -k filled > $O/filled_run$i.log 2>&1; tail -n 3 $O/filled_run$i.log; done
cat $O/filled_step_measured.jsonl
TAG=_a bash tools/micro/r06_prof_step.sh
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_a.json 2> $O/bench_driver_a.err; tail -c 600 $O/bench_driver_a.json
