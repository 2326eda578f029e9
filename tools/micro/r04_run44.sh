R=$GRAFT_REPO_ROOT
cd $R
timeout 3000 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -90 | cut -c1-400
