R=$GRAFT_REPO_ROOT
cd $R
for b in 1 2 4 8; do echo "GRIT_MSDA_FWD_BATCH=$b"; GRIT_MSDA_FWD_BATCH=$b timeout 300 python tools/micro/msda_spread.py 2>&1 | grep bf16; done
