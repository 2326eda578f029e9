R=$GRAFT_REPO_ROOT
cd $R
ITERS=60 timeout 900 python tools/micro/stress_attn_flake.py 2>&1 | grep -v Warn | tail -15
