#!/bin/bash
# A/B of one environment knob on the training step, alternating on one box:  ab_env.sh VAR v1 v2 ...   (two passes)
VAR=$1; shift
OUT=${GRIT_AB_OUT:-gpurun_out/r03}
mkdir -p $OUT
for pass in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout 400 python bench.py --no-cpu-baseline --no-analysis --steps 40 --warmup 15 2>/dev/null | tail -1 \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms', 'loss', round(d.get('final_loss',0),4))"
  done
done | tee $OUT/ab_$VAR.txt
