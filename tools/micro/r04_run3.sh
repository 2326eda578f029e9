R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 tools/micro/bin/gemm_ps_bench > $O/gemm_ps_bench_2.txt 2>&1; cat $O/gemm_ps_bench_2.txt
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_attn_gpu.py -q -x 2>&1 | tail -2; done
