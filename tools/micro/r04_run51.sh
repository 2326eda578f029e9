R=$GRAFT_REPO_ROOT
cd $R
export GRIT_AB_OUT=gpurun_out/r04
run() { # label, env
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],2))"
}
for pass in 1 2; do
run GRIT_X=0
run GRIT_WGRAD_TN_PAIR=0
run GRIT_WGRAD_PARK=0
run GRIT_SLAB_DEFER_LONG=1
run GRIT_WGRAD_DEFER_LONG=1
done
