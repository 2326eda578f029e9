#!/bin/bash
# same-box A/B: independent branches of the captured step forked onto side streams (GRIT_STEP_FORK) on / off, alternating passes
out=gpurun_out/r05_ab_fork.txt; : > $out
for pass in 1 2; do
  for f in 0 1; do
    GRIT_STEP_FORK=$f python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fork=$f pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'), d['config'].get('step_graph_error'))" >> $out
  done
done
cat $out
