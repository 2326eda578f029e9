R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python tools/micro/bench_w4_vs_lib.py 2>&1 | grep -v Warn | tee $O/w4_v0_vs_lib.txt | cut -c1-250
