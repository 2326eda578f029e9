R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_stream_kernels_gpu.py -x -q -k "patch_embedding or patch_merging" 2>&1 | tail -12 | cut -c1-250
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_abi.py -x -q 2>&1 | tail -3
