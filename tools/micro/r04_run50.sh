R=$GRAFT_REPO_ROOT
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -v Warn | tail -4
