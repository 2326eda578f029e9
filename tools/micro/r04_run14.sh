R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python tools/micro/bench_s0_gemms.py 2>&1 | grep -v Warn | tee $O/s0_gemms.txt | cut -c1-250
timeout 1200 python -m pytest tests/test_msda_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -4
