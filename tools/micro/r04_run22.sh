R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for v in 15 0 13 7 15 0; do
GRIT_GEMM_NT_AUX=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_nt_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_nt_$v.json').read().strip().splitlines()[-1]);print('GRIT_GEMM_NT_AUX=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done
