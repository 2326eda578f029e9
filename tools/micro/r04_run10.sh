R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python -X faulthandler bench.py --steps 6 --warmup 4 --no-cpu-baseline > $O/bench_fh.json 2> $O/bench_fh.err
echo rc=$?
grep -v Warning $O/bench_fh.err | tail -40 | cut -c1-200
for i in 1 2; do timeout 600 python -m pytest tests/test_graph_step_gpu.py -x -q 2>&1 | grep -E "eager|replayed|passed|failed|^E  " | cut -c1-400 | head -12; done
