#!/bin/bash
# 300 training steps on one box, second half of round 6: the new paths (short-map own tiles, own stacked value projection, d(pad) in the bias
# sum + one zero fill, residual epilogue on narrow outputs) against the first half's tree; graph replay, eager launches, one-rank RCCL
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
OFF="GRIT_GEMM_OWN_SHORT=0 GRIT_GEMM_OWN_MAX_TILES=8192 GRIT_DET_VALUE_DGRAD_NT=0 GRIT_WINATTN_PAD_VIA_BIAS=0 GRIT_WINATTN_ZERO_ARENA=0 GRIT_GEMM_RESIDUAL_NARROW=0 GRIT_FFN_RELU_EPILOGUE=0 GRIT_GROUPED_REL_BIAS=0 GRIT_GROUPED_REL_BIAS_BWD=0"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-analysis > $O/soak2_$name.json 2>/dev/null
  python3 -c "
import json;d=json.loads(open('$O/soak2_$name.json').read().strip().splitlines()[-1]);print('%-44s %6.1f img/s %6.2f ms/step  loss after 305 steps %.4f  graph %s segments %s reason %s' % ('$name', d['value'], d['ms_per_step'], d['final_loss'], d['config']['step_graph'], d['config'].get('step_graph_segments'), d['config'].get('step_graph_reason')))"
}
run graph_step_new_paths GRIT_X=0
run graph_step_first_half_paths $OFF
run eager_launches_new_paths GRIT_STEP_GRAPH=0
run rccl_one_rank_eager_default GRIT_BENCH_SELF_COLLECTIVES=1
run rccl_one_rank_segments_opt_in GRIT_BENCH_SELF_COLLECTIVES=1 GRIT_STEP_GRAPH_SEGMENTS=1
run graph_step_seed1 GRIT_BENCH_SEED=1
run graph_step_seed1_first_half_paths GRIT_BENCH_SEED=1 $OFF
