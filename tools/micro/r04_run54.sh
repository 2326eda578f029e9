R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for v in 0 1 0 1; do
GRIT_WINATTN_BWD_SKIP=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_waskip_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_waskip_$v.json').read().strip().splitlines()[-1]);print('WINATTN_BWD_SKIP=$v', round(d['value'],1), round(d['ms_per_step'],2), d['final_loss'])"
done
