R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python tools/op_profile.py --steps 2 > $O/op_profile.txt 2>&1
grep -n "819200\|204800" $O/op_profile.txt | head -30 | cut -c1-200
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_frozen_own_$i.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_frozen_own_$i.json').read().strip().splitlines()[-1]);print('own=1', round(d['value'],1), round(d['ms_per_step'],2))"
GRIT_GEMM_OWN=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_frozen_own0_$i.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_frozen_own0_$i.json').read().strip().splitlines()[-1]);print('own=0', round(d['value'],1), round(d['ms_per_step'],2))"
done
