# whole GPU suite on the current tree
R=$GRAFT_REPO_ROOT
cd $R
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
