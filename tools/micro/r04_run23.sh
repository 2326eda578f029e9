R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for v in 0 8 0 8; do
GRIT_WGRAD_TN_DBG=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_ntpart_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_ntpart_$v.json').read().strip().splitlines()[-1]);print('WGRAD_TN_DBG=$v (8 = NT partial stores)', round(d['value'],1), round(d['ms_per_step'],2))"
done
