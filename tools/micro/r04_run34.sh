R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 120 tools/micro/bin/mfma_peak | tee $O/mfma_peak.txt
