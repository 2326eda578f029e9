R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_bench_launcher.py tests/test_configs_gpu.py tests/test_ddp_gloo.py tests/test_graph_step_gpu.py -x -q -m gpu 2>&1 | tail -60 | cut -c1-300
