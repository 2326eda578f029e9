R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
GRIT_BENCH_SELF_COLLECTIVES=1 GRIT_STEP_GRAPH_COLLECTIVES=1 timeout 600 python -X faulthandler bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_rccl_one_rank_graph.json 2> $O/bench_rccl_graph.err
echo rc=$?
grep -v Warn $O/bench_rccl_graph.err | tail -8 | cut -c1-300
python3 -c "
import json;d=json.loads(open('$O/bench_rccl_one_rank_graph.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['config']['step_graph'],d['config']['step_graph_error'],d['config']['grad_allreduce'][:80], d['final_loss'])"
GRIT_BENCH_SELF_COLLECTIVES=1 timeout 600 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_rccl_one_rank_eager.json 2>/dev/null
python3 -c "
import json;d=json.loads(open('$O/bench_rccl_one_rank_eager.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['config']['step_graph'],d['final_loss'])"
