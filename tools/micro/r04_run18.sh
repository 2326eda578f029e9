R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "--- normal"
SHAPES=51200x2048x512,51200x512x2048,12800x4096x1024,204800x1024x256 timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | cut -c1-230
echo "--- every workgroup of a slice loads tile (0,0) (wrong results; traffic diagnostic)"
GRIT_WGRAD_TN_DBG=1 SHAPES=51200x2048x512,51200x512x2048,12800x4096x1024,204800x1024x256 timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | cut -c1-230
