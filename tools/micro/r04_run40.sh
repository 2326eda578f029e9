R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python tools/micro/bench_saved_dgelu.py 2>&1 | grep -v Warn | tee $O/saved_dgelu.txt | cut -c1-220
