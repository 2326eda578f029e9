# The bench lines profiles/r02 records next to the headline (run on the GPU box from the repo root).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout 240 python bench.py > $O/bench_default.json 2> $O/bench_default.err
GRIT_MSDA_BWD_F32ACC=1 timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > $O/bench_msda_f32acc.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --points spread > $O/bench_points_spread.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --ragged > $O/bench_ragged.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 6 --warmup 3 --fp32 > $O/bench_fp32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/default_stats -- python3 $R/bench.py --no-cpu-baseline --no-analysis > $O/bench_default_under_rocprof.json 2>/dev/null
cp /tmp/default_stats/*/*_kernel_stats.csv $O/bench_default_command_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-analysis > /dev/null 2>&1
python3 $R/tools/steady_profile.py /tmp/steady > $O/bench_bs32_steady_state.txt 2>&1
# HBM traffic counters: tools/micro/r02_pmc.sh (separate --pmc passes)
for f in $O/*.json; do echo "== $f"; tail -1 $f | cut -c1-200; done
