# four-wave wgrad_tn: parity tests, then the micro-benchmark against the eight-wave kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "weight_grad or wgrad or grouped_long" 2>&1 | tail -8
echo "--- eight-wave (GRIT_WGRAD_TN_W4=0)"
GRIT_WGRAD_TN_W4=0 timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | tee $O/wgrad_tn_8wave.txt | cut -c1-260
echo "--- four-wave"
timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | tee $O/wgrad_tn_4wave.txt | cut -c1-260
