#!/bin/bash
# The N > 1 code path on the ONE GPU of the box (plumbing evidence, not scaling numbers): two gloo ranks sharing the GPU, the round-6
# default (eager launches) and the opt-in segmented capture; the torchrun form the driver uses for N > 1
O=gpurun_out/r06; mkdir -p $O
GRIT_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 3 --no-analysis --no-cpu-baseline > $O/bench_gloo_2ranks_eager_default.json 2> $O/bench_gloo_2ranks_eager_default.err
GRIT_BENCH_BACKEND=gloo GRIT_STEP_GRAPH_SEGMENTS=1 python bench.py --gpus 2 --steps 6 --warmup 3 --no-analysis --no-cpu-baseline > $O/bench_gloo_2ranks_segments_opt_in.json 2> $O/bench_gloo_2ranks_segments_opt_in.err
GRIT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 3 --no-analysis --no-cpu-baseline > $O/bench_gloo_2ranks_torchrun.json 2> $O/bench_gloo_2ranks_torchrun.err
for f in bench_gloo_2ranks_eager_default bench_gloo_2ranks_segments_opt_in bench_gloo_2ranks_torchrun; do
  python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2),'ms', round(d['value'],1),'img/s n_gpus',d['n_gpus'],'graph',c.get('step_graph'),'segments',c.get('step_graph_segments'),'reason',c.get('step_graph_reason'),'err',c.get('step_graph_error'),'loss',round(d['final_loss'],4))
except Exception as e:
    print(sys.argv[1], 'NO LINE', e)
PY
done
tail -3 $O/bench_gloo_2ranks_*.err | cut -c1-300
