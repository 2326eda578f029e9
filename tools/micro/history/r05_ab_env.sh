#!/bin/bash
# same-box A/B of one environment knob: tools/micro/r05_ab_env.sh VAR "values..." [passes]   (alternating passes, bench.py timed region)
VAR=$1; VALS=$2; PASSES=${3:-2}
out=gpurun_out/r05_ab_$VAR.txt; : > $out
for pass in $(seq 1 $PASSES); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'), d['config'].get('step_graph_error'))" >> $out
  done
done
cat $out
