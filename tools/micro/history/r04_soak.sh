# 300 training steps on one box: the round's new paths (graph step, four-wave kernels, own GEMMs) against the paths they replaced,
# and the spread between two seeds (other initial weights and dropout masks)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-analysis > $O/soak_$name.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('$O/soak_$name.json').read().strip().splitlines()[-1]);print('%-44s %6.1f img/s %6.2f ms/step  loss after 305 steps %.4f  graph %s' % ('$name', d['value'], d['ms_per_step'], d['final_loss'], d['config']['step_graph']))"
}
run graph_step_new_kernels GRIT_X=0
run eager_launches_new_kernels GRIT_STEP_GRAPH=0
run graph_step_replaced_paths GRIT_GEMM_OWN=0 GRIT_WGRAD_TN_W4=0
run eager_replaced_paths GRIT_STEP_GRAPH=0 GRIT_GEMM_OWN=0 GRIT_WGRAD_TN_W4=0
run graph_step_seed1 GRIT_BENCH_SEED=1
run eager_launches_seed1 GRIT_STEP_GRAPH=0 GRIT_BENCH_SEED=1
run graph_step_seed2 GRIT_BENCH_SEED=2
run eager_launches_seed2 GRIT_STEP_GRAPH=0 GRIT_BENCH_SEED=2
