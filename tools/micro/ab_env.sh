#!/bin/bash
# same-box A/B of one environment knob on the training step: tools/micro/ab_env.sh VAR "v1 v2 ..." [passes] [extra bench args]
# (alternating passes, bench.py's timed region; results appended to gpurun_out/ab_VAR.txt)
VAR=$1; VALS=$2; PASSES=${3:-2}; EXTRA=${4:-}
mkdir -p gpurun_out; out=gpurun_out/ab_$VAR.txt; : > $out
for pass in $(seq 1 $PASSES); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'))" >> $out
  done
done
cat $out
