# end of round 4: the whole GPU suite, then the bench lines / profiles / PMC summary of the final tree
R=$GRAFT_REPO_ROOT
cd $R
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6
bash tools/micro/r04_bench_lines.sh 2>&1 | grep -v '^"' | tail -30
bash tools/micro/r04_pmc.sh 2>&1 | grep "family\|rc=" | cut -c1-200
