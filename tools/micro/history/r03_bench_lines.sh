# The bench lines and profiles profiles/r03 records next to the headline (run on the GPU box from the repo root).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
GRIT_MSDA_BWD_F32ACC=0 timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > $O/bench_msda_bf16acc.json 2>/dev/null
GRIT_MSDA_BWD_METHOD=staged timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > $O/bench_msda_staged.json 2>/dev/null
GRIT_MSDA_BWD_METHOD=staged timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --points spread > $O/bench_points_spread_staged.json 2>/dev/null
GRIT_BENCH_SELF_COLLECTIVES=1 timeout 240 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_rccl_one_rank_allreduce.json 2>/dev/null
GRIT_BENCH_SELF_COLLECTIVES=1 GRIT_GRAD_SYNC=shard timeout 240 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_rccl_one_rank_shard.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --points spread > $O/bench_points_spread.json 2>/dev/null
GRIT_MSDA_BWD_F32ACC=0 timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --points spread > $O/bench_points_spread_bf16acc.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --ragged > $O/bench_ragged.json 2>/dev/null
GRIT_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --no-cpu-baseline --no-analysis --steps 10 --warmup 4 > $O/bench_gloo_2ranks_sharing_one_gpu.json 2> $O/bench_gloo.err
GRIT_BENCH_BACKEND=gloo GRIT_GRAD_SYNC=shard timeout 400 python bench.py --gpus 2 --no-cpu-baseline --no-analysis --steps 10 --warmup 4 > $O/bench_gloo_2ranks_sharded_optimizer.json 2>> $O/bench_gloo.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/default_stats -- python3 $R/bench.py --no-cpu-baseline --no-analysis > $O/bench_default_under_rocprof.json 2>/dev/null
cp /tmp/default_stats/*/*_kernel_stats.csv $O/bench_default_command_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-analysis > /dev/null 2>&1
python3 $R/tools/steady_profile.py /tmp/steady > $O/bench_bs32_steady_state.txt 2>&1
cd $R
for f in $O/bench_*.json; do echo "== $f"; grep '^{' $f | tail -1 | cut -c1-160; done
head -3 $O/bench_bs32_steady_state.txt
