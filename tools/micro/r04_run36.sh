R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_graph_step_gpu.py -x -q 2>&1 | tail -4
