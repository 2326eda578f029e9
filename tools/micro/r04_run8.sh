R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_configs_gpu.py -q -k "a14 or config5" > $O/test_configs.txt 2>&1; tail -30 $O/test_configs.txt
