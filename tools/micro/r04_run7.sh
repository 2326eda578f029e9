R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_graph_step_gpu.py -x -q > $O/test_gemm_graph.txt 2>&1; tail -12 $O/test_gemm_graph.txt
for own in 1 0 1 0; do
GRIT_GEMM_OWN=$own timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GRIT_GEMM_OWN=$own', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('step_graph'), d['final_loss'])"
done | tee $O/ab_GRIT_GEMM_OWN.txt
