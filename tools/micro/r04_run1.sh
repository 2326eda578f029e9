# round 4, first GPU call: the graph-captured step (tests + driver command, graph vs eager on one box), then the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_graph_step_gpu.py -x -q > $O/test_graph_step.txt 2>&1; tail -15 $O/test_graph_step.txt
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_graph_driver_cmd.json 2> $O/bench_graph_driver_cmd.err
GRIT_STEP_GRAPH=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-analysis > $O/bench_eager_driver_cmd.json 2> $O/bench_eager.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-analysis > $O/bench_graph_2.json 2>> $O/bench_graph_driver_cmd.err
GRIT_STEP_GRAPH=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-analysis > $O/bench_eager_2.json 2>> $O/bench_eager.err
for f in $O/bench_*.json; do echo "== $f"; grep '^{' $f | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('step_graph'), d['config'].get('step_graph_error'))"; done
tail -5 $O/bench_graph_driver_cmd.err
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 300 python tools/bench_gemm.py 4 5 > $O/bench_gemm_v4_v5.txt 2>&1; cat $O/bench_gemm_v4_v5.txt
