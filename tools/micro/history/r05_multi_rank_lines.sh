#!/bin/bash
# The N > 1 code path on the ONE GPU of the box (plumbing evidence, not scaling numbers): one-rank RCCL self-collectives and two gloo
# ranks sharing the GPU, each with the step captured in segments around the bucket all-reduces and with eager launches.
O=gpurun_out/r05; mkdir -p $O
A="--steps 30 --warmup 8 --no-analysis --no-cpu-baseline"
GRIT_BENCH_SELF_COLLECTIVES=1 python bench.py $A > $O/bench_rccl_one_rank_segments.json 2> $O/bench_rccl_one_rank_segments.err
GRIT_BENCH_SELF_COLLECTIVES=1 GRIT_STEP_GRAPH=0 python bench.py $A > $O/bench_rccl_one_rank_eager.json 2> $O/bench_rccl_one_rank_eager.err
python bench.py $A > $O/bench_no_group_graph.json 2> /dev/null
GRIT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 3 --no-analysis --no-cpu-baseline > $O/bench_gloo_2ranks_segments.json 2> $O/bench_gloo_2ranks_segments.err
GRIT_BENCH_BACKEND=gloo GRIT_STEP_GRAPH=0 python bench.py --gpus 2 --steps 6 --warmup 3 --no-analysis --no-cpu-baseline > $O/bench_gloo_2ranks_eager.json 2> $O/bench_gloo_2ranks_eager.err
for f in bench_rccl_one_rank_segments bench_rccl_one_rank_eager bench_no_group_graph bench_gloo_2ranks_segments bench_gloo_2ranks_eager; do
  python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2),'ms', round(d['value'],1),'img/s n_gpus',d['n_gpus'],'graph',c.get('step_graph'),'segments',c.get('step_graph_segments'),'err',c.get('step_graph_error'),'loss',round(d['final_loss'],4))
except Exception as e:
    print(sys.argv[1], 'NO LINE', e)
PY
done
tail -3 $O/*.err | cut -c1-300
