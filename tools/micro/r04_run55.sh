R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "dropped or fused_mlp or epilogues" 2>&1 | tail -5 | cut -c1-250
for v in 0 1 0 1; do
GRIT_GEMM_ROW_SKIP=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_rowskip2_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_rowskip2_$v.json').read().strip().splitlines()[-1]);print('ROW_SKIP=$v (fwd + bwd)', round(d['value'],1), round(d['ms_per_step'],2), d['final_loss'])"
done
