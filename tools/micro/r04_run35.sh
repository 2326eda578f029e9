R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
GRIT_WGRAD_TN_DBG=16 SHAPES=51200x2048x512,12800x4096x1024,204800x1024x256 timeout 300 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M\|counter" | cut -c1-200 | tee $O/wgrad_tn_cycles.txt
GRIT_WGRAD_TN_DBG=17 SHAPES=51200x2048x512 timeout 300 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M\|counter" | cut -c1-200 | tee -a $O/wgrad_tn_cycles.txt
