R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python tools/micro/bench_w4_vs_lib.py > $O/w4_vs_lib.txt 2>&1; cat $O/w4_vs_lib.txt
