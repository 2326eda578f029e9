R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do timeout 900 python -m pytest tests/test_graph_step_gpu.py -x -q -k rccl 2>&1 | tail -40 | cut -c1-300; done
