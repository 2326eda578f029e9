#!/bin/bash
# round 6: in-step A/B of the own deep-K GEMMs (224-row tiles) and the residual epilogue; alternating passes on one box
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q > $O/w4_tests2.log 2>&1; tail -n 5 $O/w4_tests2.log
out=$O/ab_own_deep.txt; : > $out
for pass in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    GRIT_GEMM_OWN_DEEP=$1 GRIT_GEMM_RESIDUAL=$2 timeout 300 python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OWN_DEEP=$1 RESIDUAL=$2 pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'))" >> $out
  done
done
cat $out
