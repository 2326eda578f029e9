R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
GRIT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-analysis --steps 6 --warmup 3 > $O/bench_gloo_2ranks_sharing_one_gpu.json 2> $O/bench_gloo.err
echo rc=$?
tail -3 $O/bench_gloo.err | cut -c1-300
python3 -c "
import json;d=json.loads(open('$O/bench_gloo_2ranks_sharing_one_gpu.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['n_gpus'],d['config']['step_graph'],d['config']['grad_allreduce'][:60])"
GRIT_BENCH_BACKEND=gloo GRIT_GRAD_SYNC=shard timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-analysis --steps 6 --warmup 3 > $O/bench_gloo_2ranks_sharded_optimizer.json 2>> $O/bench_gloo.err
echo rc=$?
python3 -c "
import json;d=json.loads(open('$O/bench_gloo_2ranks_sharded_optimizer.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['n_gpus'],d['config']['grad_sync'])"
python bench.py --gpus 2 --steps 2 --warmup 1; echo "launcher on a 1-GPU box: rc=$?"
