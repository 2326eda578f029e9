# round 4, second GPU call: the persistent stream GEMM stand-alone (correctness + time + phase stamps), then the GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 tools/micro/bin/gemm_ps_bench > $O/gemm_ps_bench_1.txt 2>&1; cat $O/gemm_ps_bench_1.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
