# The bench lines and profiles profiles/r04 records next to the headline (run on the GPU box from the repo root).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
GRIT_STEP_GRAPH=0 timeout 300 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_eager_launches.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_graph_step.json 2>/dev/null
GRIT_GEMM_OWN=0 timeout 300 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_gemm_own_off.json 2>/dev/null
GRIT_WGRAD_TN_W4=0 timeout 300 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_wgrad_eight_wave.json 2>/dev/null
GRIT_BENCH_SELF_COLLECTIVES=1 timeout 300 python bench.py --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_rccl_one_rank_allreduce.json 2>/dev/null
timeout 300 python bench.py --batch 16 --no-cpu-baseline --no-analysis --steps 30 --warmup 10 > $O/bench_bs16.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/default_stats -- python3 $R/bench.py --no-cpu-baseline --no-analysis > $O/bench_default_under_rocprof.json 2>/dev/null
cp /tmp/default_stats/*/*_kernel_stats.csv $O/bench_default_command_kernel_stats.csv
rm -rf /tmp/steady; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-analysis > /dev/null 2>&1
python3 $R/tools/steady_profile.py /tmp/steady > $O/bench_bs32_steady_state.txt 2>&1
rm -rf /tmp/steady16; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady16 -- python3 $R/bench.py --batch 16 --steps 6 --warmup 4 --no-cpu-baseline --no-analysis > /dev/null 2>&1
python3 $R/tools/steady_profile.py /tmp/steady16 > $O/bench_bs16_steady_state.txt 2>&1
cd $R
for f in $O/bench_*.json; do echo "== $f"; grep '^{' $f | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('step_graph'), (d.get('config3_bs16') or {}).get('images_per_sec'))"; done
head -2 $O/bench_bs32_steady_state.txt; head -2 $O/bench_bs16_steady_state.txt
grep "decoder phase" $O/bench_bs32_steady_state.txt $O/bench_bs16_steady_state.txt | cut -c1-200
head -12 $O/bench_default_command_kernel_stats.csv | cut -c1-200
