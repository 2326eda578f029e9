R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python tools/micro/bench_fused_variants.py 2>&1 | grep -v Warn | tee $O/fused_variants.txt | cut -c1-250
