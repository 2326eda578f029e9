R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_stream_kernels_gpu.py -x -q -k "weight_grad or wgrad or grouped or tn or defer or park" 2>&1 | tail -4
for v in 4 0 4 0; do
GRIT_WGRAD_TN_DBG=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_xcdgroup_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_xcdgroup_$v.json').read().strip().splitlines()[-1]);print('DBG=$v (4 = block order)', round(d['value'],1), round(d['ms_per_step'],2))"
done
