#!/bin/bash
# round 6, run 2: tolerance measurement of test_a14 (3 runs) + NT-policy sweep of the fused Mlp GEMMs in the step (same box, alternating passes)
O=gpurun_out/r06; mkdir -p $O
rm -f $O/a14_measured.jsonl
for i in 1 2 3; do
  GRIT_TEST_MEASURE=$O/a14_measured.jsonl timeout 600 python -m pytest tests/test_configs_gpu.py -x -q -k a14 > $O/a14_run$i.log 2>&1
  tail -n 2 $O/a14_run$i.log
done
cat $O/a14_measured.jsonl
timeout 900 python -m pytest tests/test_graph_step_gpu.py -x -q > $O/graph_step_tests.log 2>&1; tail -n 3 $O/graph_step_tests.log
out=$O/ab_nt_aux.txt; : > $out
for pass in 1 2; do
  for v in 15 13 14 12 7 5; do
    GRIT_GEMM_NT_AUX=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GRIT_GEMM_NT_AUX=$v pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'))" >> $out
  done
done
cat $out
