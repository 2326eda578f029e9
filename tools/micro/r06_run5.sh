O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_stream_kernels_gpu.py -x -q -k "packed or deferred" > $O/packed_tests.log 2>&1; tail -n 15 $O/packed_tests.log
timeout 1200 python -m pytest tests/test_det_rows.py tests/test_graph_step_gpu.py -x -q > $O/det_tests.log 2>&1; tail -n 5 $O/det_tests.log
out=$O/ab_packed_in_proj.txt; : > $out
for pass in 1 2; do
  for v in 0 1; do
    GRIT_DET_PACKED_IN_PROJ=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-analysis --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PACKED_IN_PROJ=$v pass=$pass', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'img/s', 'loss', round(d['final_loss'],4), d['config'].get('step_graph'))" >> $out
  done
done
cat $out
