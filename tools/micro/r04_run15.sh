R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "narrow or policy" 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_narrow_$i.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_narrow_$i.json').read().strip().splitlines()[-1]);print('narrow own', round(d['value'],1), round(d['ms_per_step'],2))"
done
