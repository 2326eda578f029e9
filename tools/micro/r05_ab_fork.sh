#!/bin/bash
# same-box A/B: independent branches of the captured step forked onto side streams; GRIT_STEP_FORK_MASK bit 0 = grid net beside the
# detection module, bit 1 = region cross-attention beside grid cross-attention; alternating passes
out=gpurun_out/r05_ab_fork.txt; : > $out
for pass in 1 2; do
  for m in ${MASKS:-0 1 2 3}; do
    GRIT_STEP_FORK_MASK=$m python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask=$m pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'), d['config'].get('step_graph_error'))" >> $out
  done
done
cat $out
