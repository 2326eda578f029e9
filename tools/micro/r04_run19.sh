R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "weight_grad or wgrad or grouped_long or tn" 2>&1 | tail -4
timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | tee $O/wgrad_tn_4wave.txt | cut -c1-230
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_csmfma_$i.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_csmfma_$i.json').read().strip().splitlines()[-1]);print('cs-mfma', round(d['value'],1), round(d['ms_per_step'],2))"
done
